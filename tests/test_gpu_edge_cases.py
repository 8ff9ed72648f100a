"""Edge cases of the path, HIP vs oracle (bit-exact per-line state / histograms unless noted):
ragged and empty line sets, objects leaving the image or behind the camera, adaptive coverage,
non-default distribution / function lengths, every template and the generic pixel-walk path,
heterogeneous parameters inside one batch, several bodies sharing one camera (YCB shape)."""
import numpy as np
import pytest

import scenes
import util
from test_gpu_parity import _assert_lines_equal
from util import host, syn

pytestmark = pytest.mark.gpu


def _both(inputs, **kw):
    hip, ora = util.open_hip(), util.open_oracle()
    hip.call("set_fused_step", 0)
    a = scenes.Instance(hip, inputs, **kw)
    b = scenes.Instance(ora, inputs, **kw)
    for inst in (a, b):
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
    return a, b


def _compare_step(a, b, it=0, n_corr=None):
    n_corr = n_corr or a.tracker.n_corr_iterations
    for c in range(n_corr):
        a.set_poses(b.poses())
        assert a.tracker.CalculateCorrespondences(it, c) and b.tracker.CalculateCorrespondences(it, c)
        for ra, rb in zip(a.region, b.region):
            _assert_lines_equal(ra.data_lines(), rb.data_lines())
        for u in range(2):
            a.set_poses(b.poses())
            assert a.tracker.CalculateGradientAndHessian(it, c, u) and b.tracker.CalculateGradientAndHessian(it, c, u)
            for ra, rb in zip(a.region, b.region):
                ga, ha = ra.gradient_hessian()
                gb, hb = rb.gradient_hessian()
                assert np.array_equal(ga, gb) and np.array_equal(ha, hb)
            assert a.tracker.CalculateOptimization(it, c, u) and b.tracker.CalculateOptimization(it, c, u)
            assert np.array_equal(np.stack(a.poses()), np.stack(b.poses()))
    assert a.tracker.CalculateResults(it) and b.tracker.CalculateResults(it)


@pytest.mark.parametrize("scales", [[1], [3, 2], [4, 3], [6, 1], [7, 4, 2], [8, 5], [9, 7, 5, 2], [10, 3], [12]])
def test_every_scale_variant(scales):
    """region_segments<1..9> and the generic path (scale >= 10): lines bit-exact"""
    inputs = scenes.Inputs(2, 1, n_divides=2)
    rp = dict(syn.RBOT_REGION_PARAMS, scales=scales, standard_deviations=[15.0] * len(scales))
    a, b = _both(inputs, region_params=rp)
    _compare_step(a, b, n_corr=len(scales))


@pytest.mark.parametrize("fl,dl", [(8, 12), (6, 10), (4, 16), (10, 8), (16, 16), (1, 2)])
def test_function_and_distribution_lengths(fl, dl):
    inputs = scenes.Inputs(2, 1, n_divides=2)
    rp = dict(syn.RBOT_REGION_PARAMS, function_length=fl, distribution_length=dl, scales=[2, 1],
              standard_deviations=[7.0, 1.5])
    a, b = _both(inputs, region_params=rp)
    _compare_step(a, b, n_corr=2)


def test_ragged_and_empty_line_sets():
    """object half outside the image, fully outside, and behind the camera: fewer / zero valid lines,
    zero g/H leaves the pose untouched, histograms stay uniform when nothing is sampled"""
    inputs = scenes.Inputs(3, 1, n_divides=2)
    W = inputs.intr["width"]
    z = inputs.gt[0][0][2, 3]
    # object 0: centre on the right image border; object 1: far outside; object 2: behind the camera
    inputs.start[0] = inputs.gt[0][0].copy()
    inputs.start[0][0, 3] = (W - 1 - inputs.intr["ppu"]) * z / inputs.intr["fu"]
    inputs.start[1] = inputs.gt[1][0].copy()
    inputs.start[1][0, 3] += 3.0
    inputs.start[2] = inputs.gt[2][0].copy()
    inputs.start[2][2, 3] = -0.5
    a, b = _both(inputs)
    fa, ba = a.region[1].histograms()
    assert np.all(fa == np.float32(1.0) / np.float32(32 ** 3)) and np.array_equal(fa, b.region[1].histograms()[0])
    before = b.poses()
    _compare_step(a, b, n_corr=3)
    counts = [len(r.data_lines()) for r in b.region]
    assert 0 < counts[0] < 150 and counts[1] == 0 and counts[2] == 0
    after_h, after_o = a.poses(), b.poses()
    for i in (1, 2):
        assert np.array_equal(after_o[i], before[i]) and np.array_equal(after_h[i], before[i])


def test_fewer_model_points_than_lines_and_adaptive_coverage():
    """n_lines_max larger than the model's points (clamped with a warning in the reference,
    region_modality.cpp:426-430) and use_adaptive_coverage with both reference settings"""
    inputs = scenes.Inputs(2, 1, n_divides=2, n_points=120)
    for extra in (dict(n_lines_max=200), dict(n_lines_max=100, use_adaptive_coverage=1),
                  dict(n_lines_max=100, use_adaptive_coverage=1, reference_contour_length=0.35)):
        rp = dict(syn.RBOT_REGION_PARAMS, **extra)
        a, b = _both(inputs, region_params=rp)
        _compare_step(a, b, n_corr=2)
        assert len(b.region[0].data_lines()) <= 120


def test_heterogeneous_batch():
    """objects with different parameters (lines, bins, scales, image sizes) in ONE context / launch"""
    hip, ora = util.open_hip(), util.open_oracle()
    hip.call("set_fused_step", 0)
    inputs_a = scenes.Inputs(2, 2, n_divides=2)
    inputs_b = scenes.Inputs(2, 2, n_divides=1, intr=dict(syn.RBOT_INTRINSICS, width=480, height=360, ppu=240, ppv=180),
                             first_object=7)
    out = []
    for api in (hip, ora):
        i1 = scenes.Instance(api, inputs_a)
        i2 = scenes.Instance(api, inputs_b, region_params=dict(
            syn.RBOT_REGION_PARAMS, n_lines_max=90, n_histogram_bins=8, scales=[4, 2, 1],
            standard_deviations=[15.0, 5.0, 1.5], function_slope=0.5, function_amplitude=0.43))
        for inst in (i1, i2):
            inst.upload_frame(0)
        assert i1.tracker.StartModalities(0)
        out.append((i1, i2))
    (a1, a2), (b1, b2) = out
    for c in range(3):
        for x, y in ((a1, b1), (a2, b2)):
            x.set_poses(y.poses())
        assert a1.tracker.CalculateCorrespondences(0, c) and b1.tracker.CalculateCorrespondences(0, c)
        for x, y in ((a1, b1), (a2, b2)):
            for ra, rb in zip(x.region, y.region):
                _assert_lines_equal(ra.data_lines(), rb.data_lines())
    # fused step over the mixed batch stays within one-step tolerance
    hip.call("set_fused_step", 1)
    for x, y in ((a1, b1), (a2, b2)):
        x.set_poses(y.poses())
    assert a1.tracker.ExecuteTrackingStep(0) and b1.tracker.ExecuteTrackingStep(0)
    st = [syn.pose_errors(p, q) for x, y in ((a1, b1), (a2, b2)) for p, q in zip(x.poses(), y.poses())]
    assert np.median([e[0] for e in st]) < 1e-5 and np.median([e[1] for e in st]) < 1e-6


def test_bodies_sharing_one_camera():
    """YCB shape: one colour + one depth camera, several bodies / modalities referencing them"""
    inputs = scenes.Inputs(3, 2, n_divides=2, with_depth=True)
    # one common frame: paste every object into the image / depth of object 0
    for k in range(inputs.n_frames):
        img, dep = inputs.color[0][k].copy(), inputs.depth[0][k].copy()
        for i in range(1, inputs.n_objects):
            mask = inputs.depth[i][k] < np.uint16(0.95 / inputs.depth_scale)
            img[mask] = inputs.color[i][k][mask]
            dep[mask] = np.minimum(dep[mask], inputs.depth[i][k][mask])
        for i in range(inputs.n_objects):
            inputs.color[i][k], inputs.depth[i][k] = img, dep
    results = []
    for api in (util.open_hip(), util.open_oracle()):
        cam = host.ColorCamera(api, **inputs.intr)
        dcam = host.DepthCamera(api, depth_scale=inputs.depth_scale, **inputs.intr)
        bodies, mods = [], []
        for i in range(inputs.n_objects):
            rm = host.RegionModel(api, data_points=inputs.region_models[i][0], orientations=inputs.region_models[i][1],
                                  contour_lengths=inputs.region_models[i][2])
            dm = host.DepthModel(api, data_points=inputs.depth_models[i][0], orientations=inputs.depth_models[i][1],
                                 surface_areas=inputs.depth_models[i][2])
            body = host.Body(api, inputs.start[i])
            r = host.RegionModality(api, body, cam, rm, depth_camera=dcam, **syn.YCB_REGION_PARAMS)
            d = host.DepthModality(api, body, dcam, dm, **syn.YCB_DEPTH_PARAMS)
            host.Optimizer(api, body=body, modalities=[r, d])
            bodies.append(body)
            mods.append((r, d))
        tr = host.Tracker(api, 4, 2)
        cam.UpdateImage(inputs.color[0][0])
        dcam.UpdateImage(inputs.depth[0][0])
        assert tr.StartModalities(0)
        for k in range(inputs.n_frames):
            cam.UpdateImage(inputs.color[0][k])
            dcam.UpdateImage(inputs.depth[0][k])
            assert tr.ExecuteTrackingStep(k)
        results.append(np.stack([b.body2world_pose() for b in bodies]))
    assert np.array_equal(results[0], results[1])  # reference summation order: bit-exact
    for i in range(inputs.n_objects):
        assert syn.pose_errors(results[0][i], inputs.gt[i][inputs.n_frames - 1])[1] < 0.03


def test_async_ingest_matches_blocking_upload():
    """double-buffered asynchronous ingest (page-locked frames, copy stream, slot recycling) gives
    the bit-identical pose sequence of the blocking Camera::UpdateImage hand-over"""
    n_frames = 7
    inputs = scenes.Inputs(3, n_frames, n_divides=2)
    poses = []
    for asynchronous in (False, True):
        hip = util.open_hip()
        inst = scenes.Instance(hip, inputs)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        seq = []
        if not asynchronous:
            for k in range(1, n_frames):
                inst.upload_frame(k)
                assert inst.tracker.ExecuteTrackingStep(k)
                seq.append(inst.poses())
        else:
            # one page-locked block per frame index, all cameras
            blocks = [np.stack([inputs.color[i][k] for i in range(inputs.n_objects)]) for k in range(n_frames)]
            for b in blocks:
                inst.tracker.register_host_buffer(b)
            for cam in inst.color_cams:
                cam.set_ring(2)
            for i, cam in enumerate(inst.color_cams):
                cam.upload_slot(1, blocks[1][i], asynchronous=True)
            for k in range(1, n_frames):
                inst.tracker.select_slot(k % 2)
                assert inst.tracker.ExecuteTrackingStep(k)
                if k + 1 < n_frames:  # overlaps step k; recycles the slot step k-1 read
                    for i, cam in enumerate(inst.color_cams):
                        cam.upload_slot((k + 1) % 2, blocks[k + 1][i], asynchronous=True)
                seq.append(None)  # poses fetched at the end: no host sync inside the loop
            inst.tracker.ingest_sync()
            seq[-1] = inst.poses()
            for b in blocks:
                inst.tracker.unregister_host_buffer(b)
        poses.append(seq)
    assert np.array_equal(poses[0][-1], poses[1][-1])


def test_batch_ingest_matches_blocking_upload():
    """m3t_hip_cameras_set_ring + m3t_hip_cameras_upload_batch_async: the frames of all cameras for one ring slot as
    ONE page-locked block and one transfer, double-buffered against the tracking steps: the pose sequence of the
    blocking Camera::UpdateImage hand-over, bit for bit; a block in another layout takes the per-camera path"""
    import ctypes as C
    n_frames = 6
    inputs = scenes.Inputs(4, n_frames, n_divides=2)
    ref = None
    for layout in ("blocking", "ring order", "padded rows"):
        hip = util.open_hip()
        inst = scenes.Instance(hip, inputs)
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        if layout == "blocking":
            for k in range(1, n_frames):
                inst.upload_frame(k)
                assert inst.tracker.ExecuteTrackingStep(k)
        else:
            h, w = inputs.color[0][0].shape[:2]
            pad = 0 if layout == "ring order" else 64
            blocks = []
            for k in range(n_frames):
                b = np.zeros((inputs.n_objects, h, w * 3 + pad), np.uint8)
                for i in range(inputs.n_objects):
                    b[i, :, :w * 3] = inputs.color[i][k].reshape(h, w * 3)
                inst.tracker.register_host_buffer(b)
                blocks.append(b)
            ids = (C.c_int * inputs.n_objects)(*[cam.id for cam in inst.color_cams])
            hip.call("cameras_set_ring", ids, inputs.n_objects, 2)

            def upload(slot, b):
                hip.call("cameras_upload_batch_async", ids, inputs.n_objects, slot, b.ctypes.data_as(C.c_void_p),
                         b.strides[0], b.strides[1])
            upload(1, blocks[1])
            for k in range(1, n_frames):
                inst.tracker.select_slot(k % 2)
                assert inst.tracker.ExecuteTrackingStep(k)
                if k + 1 < n_frames:
                    upload((k + 1) % 2, blocks[k + 1])
            inst.tracker.ingest_sync()
        poses = np.stack(inst.poses())
        if ref is None:
            ref = poses
        assert np.array_equal(poses, ref), layout


def test_body_with_two_region_modalities():
    """a body seen by two colour cameras carries two RegionModalities: Link::CalculateGradientAndHessian sums ALL
    modalities of the link (link.cpp:184-193).  The rigid fast path keeps one modality of each kind per body, so
    such a body takes the link kernels; poses equal the oracle's bit for bit"""
    inputs = scenes.Inputs(2, 3, n_divides=2)
    out = []
    for api in (util.open_hip(), util.open_oracle()):
        models = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2])
                  for m in inputs.region_models]
        bodies, cams = [], []
        for i in range(inputs.n_objects):
            body = host.Body(api, inputs.start[i])
            cam_a, cam_b = host.ColorCamera(api, **inputs.intr), host.ColorCamera(api, **inputs.intr)
            ra = host.RegionModality(api, body, cam_a, models[inputs.model_of[i]], **dict(syn.RBOT_REGION_PARAMS, measure_occlusions=0))
            rb = host.RegionModality(api, body, cam_b, models[inputs.model_of[i]],
                                     **dict(syn.RBOT_REGION_PARAMS, measure_occlusions=0, n_lines_max=120, n_histogram_bins=16))
            host.Optimizer(api, body=body, modalities=[ra, rb])
            bodies.append(body)
            cams.append((cam_a, cam_b))
        tracker = host.Tracker(api, 7, 2)
        seq = []
        for k in range(inputs.n_frames):
            for i, (ca, cb) in enumerate(cams):
                ca.UpdateImage(inputs.color[i][k])
                cb.UpdateImage(inputs.color[i][k])
            if k == 0:
                assert tracker.StartModalities(0)
            assert tracker.ExecuteTrackingStep(k)
            seq.append(np.stack([b.body2world_pose() for b in bodies]))
        out.append(np.stack(seq))
    assert np.array_equal(out[0], out[1])
    # and the second modality matters: a single-modality tracker ends elsewhere
    ref = scenes.Instance(util.open_oracle(), inputs)
    ref.upload_frame(0)
    assert ref.tracker.StartModalities(0)
    for k in range(inputs.n_frames):
        ref.upload_frame(k)
        assert ref.tracker.ExecuteTrackingStep(k)
    assert not np.array_equal(np.stack(ref.poses()), out[1][-1])


def test_shared_color_histograms():
    """three bodies whose RegionModalities share one ColorHistograms object (RTB configuration,
    region_modality.cpp:168-173, tracker.cpp:435-443,507-515): every modality adds its samples, the object is
    initialised / updated once; histograms and poses bit-identical to the oracle over three frames"""
    inputs = scenes.Inputs(3, 4, n_divides=2)
    out = []
    for api in (util.open_hip(), util.open_oracle()):
        inst = scenes.Instance(api, inputs)
        shared = host.ColorHistograms(api, n_bins=32, learning_rate_f=0.3, learning_rate_b=0.1)
        for r in inst.region[:2]:
            r.UseSharedColorHistograms(shared)  # the third keeps its private histograms
        inst.upload_frame(0)
        assert inst.tracker.StartModalities(0)
        states = [inst.region[0].histograms(), inst.region[2].histograms()]
        for k in range(1, 4):
            inst.upload_frame(k)
            assert inst.tracker.ExecuteTrackingStep(k)
        states += [inst.region[0].histograms(), inst.region[1].histograms(), inst.region[2].histograms()]
        out.append((states, inst.poses()))
    (sa, pa), (sb, pb) = out
    for (fa, ba), (fb, bb) in zip(sa, sb):
        assert np.array_equal(fa, fb) and np.array_equal(ba, bb)
    assert np.array_equal(sa[2][0], sa[3][0])          # modalities 0 and 1 read the same object
    assert not np.array_equal(sa[2][0], sa[4][0])
    assert np.array_equal(np.asarray(pa), np.asarray(pb))
