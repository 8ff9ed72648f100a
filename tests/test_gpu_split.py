"""tracking_step_split_kernel (four workgroups per object, partial g/H sums exchanged inside the launch) against
the one-workgroup-per-object kernel: the default summation order is defined per quarter of the lines, so the two
launch shapes must agree bit for bit — poses, histograms and the per-line state — for every batch size."""
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


def run(inputs, split, fused_mode, n_frames):
    if split:
        os.environ.pop("M3T_HIP_NO_SPLIT", None)
    else:
        os.environ["M3T_HIP_NO_SPLIT"] = "1"
    try:
        api = util.open_hip()
        api.call("set_fused_step", fused_mode)
        inst = scenes.Instance(api, inputs)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        poses = []
        for k in range(n_frames):
            inst.upload_frame(k)
            assert inst.tracker.ExecuteTrackingStep(k)
            poses.append(np.stack(inst.poses()))
        shape = step_shape(api)
        hists = [np.concatenate(r.histograms()) for r in inst.region]
        lines = [r.data_lines() for r in inst.region] if fused_mode == 2 else None
        gh = [np.concatenate([r.gradient(), r.hessian().reshape(-1)]) for r in inst.region] if fused_mode == 2 else None
        return np.stack(poses), hists, lines, gh, shape
    finally:
        os.environ.pop("M3T_HIP_NO_SPLIT", None)


@pytest.mark.parametrize("n_objects", [1, 3, 8])  # 8: the XCD-aware block -> (object, part) mapping
def test_split_kernel_is_bit_identical_to_one_workgroup_per_object(n_objects):
    inputs = scenes.Inputs(n_objects, 5, n_divides=2, n_models=min(n_objects, 2))
    pa, ha, la, ga, shape_a = run(inputs, True, 2, 5)
    pb, hb, lb, gb, shape_b = run(inputs, False, 2, 5)
    assert shape_a[:2] == [n_objects, 4] and shape_b[:2] == [n_objects, 1]
    assert shape_a[3] == 1 and shape_b[3] == 1  # the histogram update rides in both launches
    assert np.array_equal(pa, pb)
    for a, b in zip(ha, hb):
        assert np.array_equal(a, b)
    for a, b in zip(ga, gb):
        assert np.array_equal(a, b)
    for a, b in zip(la, lb):
        assert len(a) == len(b) and a.tobytes() == b.tobytes()
    # and it tracks: every object within 5 cm / 5 degrees of the ground truth after the last frame
    for i in range(n_objects):
        e = util.syn.pose_errors(pa[-1][i], inputs.gt[i][4])
        assert e[0] < np.deg2rad(5) and e[1] < 0.05


def test_split_kernel_is_not_used_where_it_does_not_apply():
    # Region + Depth objects, and the reference-summation-order mode, keep one workgroup per object
    inputs = scenes.Inputs(2, 2, with_depth=True)
    api = util.open_hip()
    inst = scenes.Instance(api, inputs, use_region=True, use_depth=True)
    inst.upload_frame(0)
    assert inst.tracker.StartModalities(0) and inst.tracker.ExecuteTrackingStep(0)
    assert step_shape(api)[:2] == [2, 1]
    inputs = scenes.Inputs(2, 2, n_divides=2)
    api = util.open_hip()
    api.call("set_summation_mode", 1)
    inst = scenes.Instance(api, inputs)
    inst.upload_frame(0)
    assert inst.tracker.StartModalities(0) and inst.tracker.ExecuteTrackingStep(0)
    assert step_shape(api)[:2] == [2, 1]
    api.call("set_summation_mode", 0)
    assert inst.tracker.ExecuteTrackingStep(1)
    assert step_shape(api)[:2] == [2, 4]
    api.call("set_object_split", 0)  # a process that shares its GPU
    assert inst.tracker.ExecuteTrackingStep(1)
    assert step_shape(api)[:2] == [2, 1]


def test_default_summation_order_does_not_depend_on_the_workgroup_size(monkeypatch):
    """large batches run 256-thread workgroups (two per CU): same poses and histograms as 512 threads, bit for bit"""
    inputs = scenes.Inputs(2, 4, n_divides=2)
    monkeypatch.setenv("M3T_HIP_THREADS", "256")
    pa, ha, _, _, shape_a = run(inputs, False, 1, 4)
    monkeypatch.delenv("M3T_HIP_THREADS")
    pb, hb, _, _, shape_b = run(inputs, False, 1, 4)
    assert shape_a[2] == 256 and shape_b[2] == 512 and shape_a[1] == shape_b[1] == 1
    assert np.array_equal(pa, pb)
    for a, b in zip(ha, hb):
        assert np.array_equal(a, b)
