"""Parity at the BENCHMARKED launch shapes (BASELINE.json configs[1] and configs[2] at their stated sizes):
bench.py's own inputs -- 64 objects x 2562-view models, RegionModality, RBOT parameters -- through
tracking_step_split_kernel with 256 workgroups (every CU busy), and 21 Region + Depth objects (YCB parameters)
over 8 workgroups each, against the oracle: body2world of every object after every frame, free running, bit for bit.
Stated tolerance: 0 (rotation, translation and ADD-S differences are exactly zero)."""
import ctypes as C
import os

import numpy as np
import pytest

import scenes
import util
from util import syn

pytestmark = pytest.mark.gpu


def shape_of(api):
    shape = (C.c_int * 4)()
    api.call("get_step_shape", shape)
    return list(shape)


@pytest.fixture(scope="module")
def rbot64():
    # bench.py --config rbot64: scenes.Inputs(64, n_frames, n_divides=4, n_models=CONFIGS["rbot64"]["models"]) -- 18 models
    import bench
    return scenes.Inputs(64, 6, n_divides=4, n_models=bench.CONFIGS["rbot64"]["models"])


def oracle_trajectory(inputs, use_depth=False):
    ora = util.open_oracle()
    b = scenes.Instance(ora, inputs, use_depth=use_depth)
    b.upload_frame(0)
    assert b.tracker.StartModalities(0)
    ref = []
    for k in range(inputs.n_frames):
        b.upload_frame(k)
        assert b.tracker.ExecuteTrackingStep(k)
        ref.append(np.stack(b.poses()))
    return ref, [r.histograms() for r in b.region]


def kernel_of(api):
    name = C.create_string_buffer(64)
    api.call("get_step_kernel", name, 64)
    return name.value.decode()


def hip_trajectory(inputs, use_depth=False, env=None, want_kernel=None):
    for k in ("M3T_HIP_NO_SPLIT", "M3T_HIP_SPLIT_PARTS", "M3T_HIP_THREADS", "M3T_HIP_COMPACT", "M3T_HIP_COMPACT_TABLE",
              "M3T_HIP_COMPACT_TABLE_KB", "M3T_HIP_COMPACT_WIDE", "M3T_HIP_NO_PAIR"):
        os.environ.pop(k, None)
    os.environ.update(env or {})
    try:
        api = util.open_hip()
        a = scenes.Instance(api, inputs, use_depth=use_depth)
        a.upload_frame(0)
        assert a.tracker.StartModalities(0)
        out = []
        for k in range(inputs.n_frames):
            a.upload_frame(k)
            assert a.tracker.ExecuteTrackingStep(k)
            out.append(np.stack(a.poses()))
        if want_kernel is not None:
            assert kernel_of(api) == want_kernel
        return out, [r.histograms() for r in a.region], shape_of(api)
    finally:
        for k in (env or {}):
            os.environ.pop(k, None)


def test_headline_batch_is_bit_identical_to_the_oracle(rbot64):
    ref, ref_hist = oracle_trajectory(rbot64)
    got, hist, shape = hip_trajectory(rbot64)
    assert shape == [64, 4, 512, 1]  # the launch bench.py times: 4 workgroups per object, all 256 CUs, one launch per frame
    for k in range(rbot64.n_frames):
        assert np.array_equal(got[k], ref[k]), k
    worst = [max(syn.pose_errors(got[-1][i], ref[-1][i])[j] for i in range(64)) for j in (0, 1)]
    adds = max(syn.add_s(rbot64.vertices[i], got[-1][i], ref[-1][i]) for i in range(64))
    assert worst == [0.0, 0.0] and adds == 0.0
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
    # and the batch is a meaningful one: every object is tracked (rbot_evaluator.cpp:416-433)
    for i in range(64):
        e = syn.pose_errors(got[-1][i], rbot64.gt[i][-1])
        assert e[0] < np.deg2rad(5) and e[1] < 0.05


def test_headline_batch_in_the_other_launch_shapes(rbot64):
    """one workgroup per object (512 and 256 threads) and 2 workgroups per object: the same bits"""
    ref, _ = oracle_trajectory(rbot64)
    for env, expect in (({"M3T_HIP_NO_SPLIT": "1"}, [64, 1, 512, 1]),
                        ({"M3T_HIP_NO_SPLIT": "1", "M3T_HIP_THREADS": "256"}, [64, 1, 256, 0])):
        got, _, shape = hip_trajectory(rbot64, env=env)
        assert shape == expect
        for k in range(rbot64.n_frames):
            assert np.array_equal(got[k], ref[k]), (env, k)


@pytest.mark.parametrize("env,kernel", [({}, "tracking_step_split_pair_kernel"),
                                        ({"M3T_HIP_NO_PAIR": "1"}, "tracking_step_split_kernel")])
def test_ycb_batch_is_bit_identical_to_the_oracle(env, kernel):
    """BASELINE configs[2]: 21 objects, Region + Depth fused modalities, YCB parameters (evaluate_ycb_dataset.cpp:46-76,
    108-133), 2562-view models; 8 workgroups per object (168 CUs).  Through the _pair_ kernel bodies with both modalities
    take (the lines' and the points' products side by side, their sums on two waves) and through the plain one."""
    inputs = scenes.Inputs(21, 5, n_divides=4, n_models=6, with_depth=True)
    ref, ref_hist = oracle_trajectory(inputs, use_depth=True)
    got, hist, shape = hip_trajectory(inputs, use_depth=True, env=env, want_kernel=kernel)
    assert shape[:3] == [21, 8, 512]
    for k in range(inputs.n_frames):
        assert np.array_equal(got[k], ref[k]), k
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
    adds = [syn.add_s(inputs.vertices[i], got[-1][i], inputs.gt[i][-1]) for i in range(21)]
    assert max(adds) < 0.01  # tracked: ADD-S against the ground truth below 1 cm


def test_compact_kernel_is_bit_identical_to_the_oracle(rbot64):
    """tracking_step_compact_kernel (the launch shape of batches with two objects per CU and more: 256-thread
    workgroups, a thread per correspondence line, factor rows instead of product rows), forced onto the headline
    inputs: poses after every frame and histograms equal the oracle's"""
    ref, ref_hist = oracle_trajectory(rbot64)
    got, hist, shape = hip_trajectory(rbot64, env={"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1", "M3T_HIP_COMPACT_TABLE": "0"},
                                      want_kernel="tracking_step_compact_kernel")
    assert shape[:3] == [64, 1, 256]
    for k in range(rbot64.n_frames):
        assert np.array_equal(got[k], ref[k]), k
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)


def test_compact_table_kernel_is_bit_identical_to_the_oracle(rbot64):
    """tracking_step_compact_table_kernel (round 6; the default for Region-only batches of more objects than CUs): the
    32-bin pair table compacted in LDS -- two bits per bin for the three constant pairs (0.5, 0.5) / (1, 0) / (0, 1), the
    pairs of the bins both histograms hold in bin order -- in place of the 8-byte gather per sample
    (region_modality.cpp:1575-1598).  Saturated pixels put bin 32767 -- the table's last bit -- into both histograms."""
    base = scenes.subset(rbot64, list(range(16)))
    base.color = [[f.copy() for f in frames] for frames in base.color]
    for frames in base.color:
        for f in frames:
            f[::8, ::8, :] = 255
    ref, ref_hist = oracle_trajectory(base)
    got, hist, shape = hip_trajectory(base, env={"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1"},
                                      want_kernel="tracking_step_compact_table_kernel")
    assert shape[:3] == [16, 1, 256]
    for k in range(base.n_frames):
        assert np.array_equal(got[k], ref[k]), k
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)


@pytest.mark.parametrize("n_bins,kernel", [(16, "tracking_step_compact_table_kernel"), (8, "tracking_step_compact_kernel"),
                                           (64, "tracking_step_compact_kernel")])
def test_compact_table_kernel_histogram_sizes(n_bins, kernel):
    """16 bins per channel: a table of 128 words (half the staging threads idle); 8 bins (512 bins in all: below the
    table's lower limit) and 64 bins (262 144: above what 16-bit ranks and slots hold) keep the kernel that gathers
    from the global table.  All three against the oracle, bit for bit."""
    inputs = scenes.Inputs(6, 4, n_divides=2, n_models=2)
    rp = dict(syn.RBOT_REGION_PARAMS, n_histogram_bins=n_bins, measure_occlusions=0)

    def trajectory(api):
        a = scenes.Instance(api, inputs, region_params=rp)
        a.upload_frame(0)
        assert a.tracker.StartModalities(0)
        out = []
        for k in range(inputs.n_frames):
            a.upload_frame(k)
            assert a.tracker.ExecuteTrackingStep(k)
            out.append(np.stack(a.poses()))
        return out, [r.histograms() for r in a.region]

    ref, ref_hist = trajectory(util.open_oracle())
    os.environ.update({"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1"})
    try:
        api = util.open_hip()
        got, hist = trajectory(api)
        assert kernel_of(api) == kernel
    finally:
        for k in ("M3T_HIP_COMPACT", "M3T_HIP_NO_SPLIT"):
            os.environ.pop(k, None)
    for k in range(inputs.n_frames):
        assert np.array_equal(got[k], ref[k]), k
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)


def test_compact_table_overflow_takes_the_global_table_and_tells_the_host(rbot64):
    """a table with room for 64 mixed bins only: the bins beyond it are read from the global pair table inside the same
    walk (same poses, bit for bit), the kernels report by how much they overflowed, and the host goes back to the
    kernel without the table from the next step on -- until StartModalities brings new histograms"""
    base = scenes.subset(rbot64, list(range(8)))
    ref, _ = oracle_trajectory(base)
    for k in ("M3T_HIP_NO_SPLIT", "M3T_HIP_SPLIT_PARTS", "M3T_HIP_THREADS", "M3T_HIP_COMPACT", "M3T_HIP_COMPACT_TABLE"):
        os.environ.pop(k, None)
    os.environ.update({"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1", "M3T_HIP_COMPACT_TABLE_CAP": "64"})
    try:
        api = util.open_hip()
        a = scenes.Instance(api, base)
        a.upload_frame(0)
        assert a.tracker.StartModalities(0)
        kernels = []
        for k in range(base.n_frames):
            a.upload_frame(k)
            assert a.tracker.ExecuteTrackingStep(k)
            assert np.array_equal(np.stack(a.poses()), ref[k]), k  # (the read-back also waits for the step)
            kernels.append(kernel_of(api))
        assert kernels[0] == "tracking_step_compact_table_kernel" and kernels[-1] == "tracking_step_compact_kernel", kernels
        assert a.tracker.StartModalities(base.n_frames - 1)
        assert a.tracker.ExecuteTrackingStep(base.n_frames - 1)
        assert kernel_of(api) == "tracking_step_compact_table_kernel"
    finally:
        for k in ("M3T_HIP_COMPACT", "M3T_HIP_NO_SPLIT", "M3T_HIP_COMPACT_TABLE_CAP"):
            os.environ.pop(k, None)


@pytest.mark.parametrize("wide,kernel,threads", [("0", "tracking_step_compact_kernel", 256),
                                                 ("1", "tracking_step_compact_wide_kernel", 512)])
def test_compact_kernel_region_and_depth(wide, kernel, threads):
    """the same for Region + Depth objects with measured occlusions (YCB parameters, 16 bins: the histogram update
    rides in the launch), with 256-thread workgroups and with the 512-thread ones batches with depth modalities take"""
    inputs = scenes.Inputs(21, 5, n_divides=4, n_models=6, with_depth=True)
    ref, ref_hist = oracle_trajectory(inputs, use_depth=True)
    got, hist, shape = hip_trajectory(inputs, use_depth=True,
                                      env={"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1", "M3T_HIP_COMPACT_WIDE": wide},
                                      want_kernel=kernel)
    assert shape == [21, 1, threads, 1]
    for k in range(inputs.n_frames):
        assert np.array_equal(got[k], ref[k]), k
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)


@pytest.mark.parametrize("wide,kernel,threads", [("0", "tracking_step_compact_kernel", 256),
                                                 ("1", "tracking_step_compact_wide_kernel", 512)])
def test_compact_kernels_depth_only(wide, kernel, threads):
    """objects that carry a DepthModality alone (no lines: every thread of either workgroup size scans points)"""
    inputs = scenes.Inputs(9, 4, n_divides=4, n_models=3, with_depth=True)
    ora, api_env = util.open_oracle(), {"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1", "M3T_HIP_COMPACT_WIDE": wide}
    b = scenes.Instance(ora, inputs, use_region=False, use_depth=True)
    os.environ.update(api_env)
    try:
        api = util.open_hip()
        a = scenes.Instance(api, inputs, use_region=False, use_depth=True)
        for inst in (a, b):
            inst.upload_frame(0)
            assert inst.tracker.StartModalities(0)
        for k in range(inputs.n_frames):
            for inst in (a, b):
                inst.upload_frame(k)
                assert inst.tracker.ExecuteTrackingStep(k)
            assert np.array_equal(np.stack(a.poses()), np.stack(b.poses())), k
        assert kernel_of(api) == kernel and shape_of(api)[:3] == [9, 1, threads], (kernel_of(api), shape_of(api))
    finally:
        for k in api_env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("env,kernel", [({}, "tracking_step_split_pair_kernel"),
                                        ({"M3T_HIP_NO_SPLIT": "1"}, "tracking_step_lds_pair_kernel"),
                                        ({"M3T_HIP_NO_PAIR": "1"}, "tracking_step_split_kernel")])
def test_mixed_batch_through_the_pair_kernels(env, kernel):
    """bodies with both modalities, with a RegionModality alone and with a DepthModality alone in ONE context: the _pair_
    kernels decide per body (side by side only where there are two modalities to put side by side)"""
    inputs = scenes.Inputs(9, 4, n_divides=4, n_models=3, with_depth=True)
    kinds = ["rd", "r", "d"] * 3
    ora = util.open_oracle()
    b = scenes.Instance(ora, inputs, kinds=kinds)
    os.environ.update(env)
    try:
        api = util.open_hip()
        a = scenes.Instance(api, inputs, kinds=kinds)
        for inst in (a, b):
            inst.upload_frame(0)
            assert inst.tracker.StartModalities(0)
        for k in range(inputs.n_frames):
            for inst in (a, b):
                inst.upload_frame(k)
                assert inst.tracker.ExecuteTrackingStep(k)
            assert np.array_equal(np.stack(a.poses()), np.stack(b.poses())), k
        assert kernel_of(api) == kernel, kernel_of(api)
        for ra, rb in zip(a.region, b.region):
            (fa, ba), (fb, bb) = ra.histograms(), rb.histograms()
            assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
    finally:
        for k in env:
            os.environ.pop(k, None)


@pytest.mark.parametrize("env,kernel,shape", [
    ({}, "tracking_step_compact_wide_kernel", [512, 1, 512, 1]),
    ({"M3T_HIP_COMPACT_WIDE": "0"}, "tracking_step_compact_kernel", [512, 1, 256, 1]),
    ({"M3T_HIP_COMPACT": "0"}, "tracking_step_lds_pair_kernel", [512, 1, 512, 1]),
    ({"M3T_HIP_COMPACT": "0", "M3T_HIP_NO_PAIR": "1"}, "tracking_step_lds_kernel", [512, 1, 512, 1])])
def test_synth512_batch_matches_the_oracle(env, kernel, shape):
    """BASELINE configs[3] at its stated size on one GPU: bench.py --config synth512's own inputs (64 rendered Region +
    Depth streams of 16 distinct 2562-view models, spread over 512 objects with their own cameras, frames and
    histograms), two and more workgroups per CU.  A sample of 32 objects, the last one included, against the oracle:
    body2world after every frame and the histograms, bit for bit; and every object equals the other objects that look
    at the same stream (nothing leaks between the workgroups that share a CU)."""
    base = scenes.Inputs(64, 4, n_divides=4, n_models=16, with_depth=True)
    inputs = scenes.replicate(base, 512)  # how bench.py spreads its 64 rendered streams over the batch
    got, hist, got_shape = hip_trajectory(inputs, use_depth=True, env=env, want_kernel=kernel)
    assert got_shape == shape
    sample = list(range(31)) + [511]
    ref, ref_hist = oracle_trajectory(scenes.subset(inputs, sample), use_depth=True)
    for k in range(inputs.n_frames):
        assert np.array_equal(got[k][sample], ref[k]), k
        for i in range(64, 512):
            assert np.array_equal(got[k][i], got[k][i % 64]), (k, i)
    for j, i in enumerate(sample):
        assert np.array_equal(hist[i][0], ref_hist[j][0]) and np.array_equal(hist[i][1], ref_hist[j][1])
    adds = [syn.add_s(inputs.vertices[i], got[-1][i], inputs.gt[i][-1]) for i in sample]
    assert max(adds) < 0.01  # tracked


def test_compact_kernel_counts_saturated_background_pixels():
    """32 bins, histogram update from the 16-bit sample list (tracking_step_compact_kernel): white pixels fall into bin
    0x7fff, which as a background sample must not be taken for the list's empty-slot word 0xffff (round-3 advisor).
    Frames with every 8th pixel of every 8th row saturated: foreground and background walks both meet bin 32767."""
    base = scenes.Inputs(8, 4, n_divides=2, n_models=4)
    inputs = scenes.subset(base, list(range(8)))
    inputs.color = [[f.copy() for f in frames] for frames in base.color]
    for frames in inputs.color:
        for f in frames:
            f[::8, ::8, :] = 255
    ref, ref_hist = oracle_trajectory(inputs)
    assert all(fb[1].reshape(-1)[32767] > 0 and fb[0].reshape(-1)[32767] > 0 for fb in ref_hist)
    got, hist, shape = hip_trajectory(inputs, env={"M3T_HIP_COMPACT": "1", "M3T_HIP_NO_SPLIT": "1", "M3T_HIP_COMPACT_TABLE": "0"},
                                      want_kernel="tracking_step_compact_kernel")
    assert shape == [8, 1, 256, 1]
    for (fa, ba), (fb, bb) in zip(hist, ref_hist):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
    for k in range(inputs.n_frames):
        assert np.array_equal(got[k], ref[k]), k
