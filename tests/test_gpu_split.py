"""tracking_step_split_kernel (several workgroups per object; the line / point results cross CUs once per
correspondence iteration, every workgroup then forms the sums in the reference's order and solves redundantly)
against the one-workgroup-per-object kernel and against the oracle: poses, histograms, per-line state and g/H
must agree bit for bit for every batch size and every number of workgroups per object."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
import util

pytestmark = pytest.mark.gpu


def step_shape(api):
    shape = (C.c_int * 4)()
    api.call("get_step_shape", shape)
    return list(shape)


def run(inputs, parts, fused_mode, n_frames, use_depth=False):
    os.environ.pop("M3T_HIP_NO_SPLIT", None)
    os.environ.pop("M3T_HIP_SPLIT_PARTS", None)
    if parts:
        os.environ["M3T_HIP_SPLIT_PARTS"] = str(parts)
    else:
        os.environ["M3T_HIP_NO_SPLIT"] = "1"
    try:
        api = util.open_hip()
        api.call("set_fused_step", fused_mode)
        inst = scenes.Instance(api, inputs, use_depth=use_depth)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        poses = []
        for k in range(n_frames):
            inst.upload_frame(k)
            assert inst.tracker.ExecuteTrackingStep(k)
            poses.append(np.stack(inst.poses()))
        shape = step_shape(api)
        hists = [np.concatenate(r.histograms()) for r in inst.region]
        state = None
        if fused_mode == 2:
            state = [r.data_lines().tobytes() for r in inst.region] + [d.data_points().tobytes() for d in inst.depth]
            state += [np.concatenate([m.gradient(), m.hessian().reshape(-1)]).tobytes() for m in inst.region + inst.depth]
        return np.stack(poses), hists, state, shape
    finally:
        os.environ.pop("M3T_HIP_NO_SPLIT", None)
        os.environ.pop("M3T_HIP_SPLIT_PARTS", None)


@pytest.mark.parametrize("n_objects,parts", [(1, 16), (3, 8), (8, 4), (8, 8)])  # 8: the XCD-aware block -> (object, part) map
def test_split_kernel_is_bit_identical_to_one_workgroup_per_object(n_objects, parts):
    inputs = scenes.Inputs(n_objects, 5, n_divides=2, n_models=min(n_objects, 2))
    pa, ha, sa, shape_a = run(inputs, parts, 2, 5)
    pb, hb, sb, shape_b = run(inputs, 0, 2, 5)
    assert shape_a[:2] == [n_objects, parts] and shape_b[:2] == [n_objects, 1]
    assert shape_a[3] == 1 and shape_b[3] == 1  # the histogram update rides in both launches
    assert np.array_equal(pa, pb)
    for a, b in zip(ha, hb):
        assert np.array_equal(a, b)
    assert sa == sb
    # and it tracks: every object within 5 cm / 5 degrees of the ground truth after the last frame
    for i in range(n_objects):
        e = util.syn.pose_errors(pa[-1][i], inputs.gt[i][4])
        assert e[0] < np.deg2rad(5) and e[1] < 0.05


def test_split_region_depth_state_written_back():
    """Region + Depth objects over 8 workgroups each, fused mode 2: the line / point state and g/H every workgroup
    ends up with (own part computed, other parts received) equals the one-workgroup kernel's"""
    inputs = scenes.Inputs(2, 3, n_divides=2, with_depth=True)
    pa, ha, sa, shape_a = run(inputs, 8, 2, 3, use_depth=True)
    pb, hb, sb, shape_b = run(inputs, 0, 2, 3, use_depth=True)
    assert shape_a[:2] == [2, 8] and shape_b[:2] == [2, 1]
    assert np.array_equal(pa, pb) and sa == sb
    for a, b in zip(ha, hb):
        assert np.array_equal(a, b)


def test_split_shapes_chosen_by_the_library():
    inputs = scenes.Inputs(2, 2, with_depth=True)
    api = util.open_hip()
    inst = scenes.Instance(api, inputs, use_region=True, use_depth=True)
    inst.upload_frame(0)
    assert inst.tracker.StartModalities(0) and inst.tracker.ExecuteTrackingStep(0)
    assert step_shape(api)[:2] == [2, 8]  # Region + Depth objects are split as well
    api.call("set_object_split", 4)       # at most 4 workgroups per object
    assert inst.tracker.ExecuteTrackingStep(1)
    assert step_shape(api)[:2] == [2, 4]
    api.call("set_object_split", 0)       # a process that shares its GPU
    assert inst.tracker.ExecuteTrackingStep(1)
    assert step_shape(api)[:2] == [2, 1]
    api.call("set_object_split", 1)
    assert inst.tracker.ExecuteTrackingStep(1)
    assert step_shape(api)[:2] == [2, 8]


def test_results_do_not_depend_on_the_workgroup_size(monkeypatch):
    """large batches run 256-thread workgroups (two per CU): same poses and histograms as 512 threads, bit for bit"""
    inputs = scenes.Inputs(2, 4, n_divides=2)
    monkeypatch.setenv("M3T_HIP_THREADS", "256")
    pa, ha, _, shape_a = run(inputs, 0, 1, 4)
    monkeypatch.delenv("M3T_HIP_THREADS")
    pb, hb, _, shape_b = run(inputs, 0, 1, 4)
    assert shape_a[2] == 256 and shape_b[2] == 512 and shape_a[1] == shape_b[1] == 1
    assert np.array_equal(pa, pb)
    for a, b in zip(ha, hb):
        assert np.array_equal(a, b)


def test_two_contexts_share_the_gpu():
    """Two contexts (two host threads, two streams) launch split kernels on one GPU at the same time.  Their
    workgroups may not all be resident together; a step either completes with the right result or is abandoned
    cleanly: the call after it reports the error, nothing hangs, and after switching the split off the sequence
    can be repeated with the expected poses."""
    import threading
    inputs = [scenes.Inputs(32, 6, n_divides=2, n_models=2, first_object=100 * t) for t in range(2)]
    ref = []
    for t in range(2):
        p, _, _, shape = run(inputs[t], 0, 1, 6)
        ref.append(p)
    results = [None, None]

    def worker(t):
        api = util.open_hip()
        inst = scenes.Instance(api, inputs[t])
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        for cam in inst.color_cams:
            api.call("camera_set_ring", cam.id, 6)
        for k in range(6):
            for i, cam in enumerate(inst.color_cams):
                f = inputs[t].color[i][k]
                api.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
        failed = False
        for rep in range(20):  # back-to-back steps without host synchronisation in between
            inst.set_poses(inputs[t].start)
            api.call("cameras_select_slot", 0)
            ok = inst.tracker.StartModalities(0)
            for k in range(6):
                api.call("cameras_select_slot", k)
                ok = ok and inst.tracker.ExecuteTrackingStep(k)
            ok = ok and api.raw("sync") == 0
            if not ok:
                failed = True
                assert "waited in vain" in api.last_error()
                break
            assert np.array_equal(np.stack(inst.poses()), ref[t][-1])
        if failed:  # recover: no split from here on
            api.call("set_object_split", 0)
            api.raw("sync")
            inst.set_poses(inputs[t].start)
            api.call("cameras_select_slot", 0)
            assert inst.tracker.StartModalities(0)
            for k in range(6):
                api.call("cameras_select_slot", k)
                assert inst.tracker.ExecuteTrackingStep(k)
            assert np.array_equal(np.stack(inst.poses()), ref[t][-1])
        results[t] = "abandoned+recovered" if failed else "ok"

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert all(not th.is_alive() for th in threads), "a context hung"
    print("two contexts:", results)
    assert all(r is not None for r in results)
