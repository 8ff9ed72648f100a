"""Evaluator front-ends (M3T/examples/rbot_evaluator.cpp, ycb_evaluator.cpp): file readers, error metrics
and the reset-on-loss loop, on known answers and on a synthetic sequence tracked by the CPU oracle."""
import os

import numpy as np
import pytest

import scenes
import util

ev = util.pkg.evaluation


def rot(axis, angle):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * K @ K


def pose(R=np.eye(3), t=(0, 0, 0)):
    p = np.eye(4, dtype=np.float32)
    p[:3, :3] = R
    p[:3, 3] = t
    return p


def test_rbot_pose_file_and_criterion(tmp_path):
    """poses_first.txt layout (rbot_evaluator.cpp:558-585): header line, rotation row-major, translation in mm"""
    R = rot((1, 2, 3), 0.3).astype(np.float32)
    path = tmp_path / "poses_first.txt"
    with open(path, "w") as f:
        f.write("r11\tr12\t...\n")
        for i in range(4):
            f.write("\t".join("%.7g" % v for v in list(R.reshape(-1)) + [10.0 * i, -20.0, 500.0 + i]) + "\n")
    poses = ev.read_poses_rbot(str(path), n_frames=3)
    assert poses.shape == (4, 4, 4)
    assert np.allclose(poses[2, :3, :3], R, atol=1e-6)
    assert np.allclose(poses[2, :3, 3], [0.02, -0.02, 0.502], atol=1e-7) and poses[2, 3, 3] == 1
    with pytest.raises(ValueError):
        ev.read_poses_rbot(str(path), n_frames=4)
    # CalculatePoseResults :416-433: 5 cm and 5 degrees, strict comparisons
    gt = pose(R, (0.1, 0.2, 0.6))
    t_err, r_err, ok = ev.rbot_pose_result(pose(R @ rot((0, 0, 1), np.deg2rad(4.9)), (0.1, 0.2, 0.649)), gt)
    assert ok == 1.0 and t_err == pytest.approx(0.049, abs=1e-6) and r_err == pytest.approx(np.deg2rad(4.9), abs=1e-4)
    assert ev.rbot_pose_result(pose(R @ rot((0, 1, 0), np.deg2rad(5.1)), (0.1, 0.2, 0.6)), gt)[2] == 0.0
    assert ev.rbot_pose_result(pose(R, (0.1, 0.2, 0.651)), gt)[2] == 0.0


def test_ycb_metrics_known_answers():
    th = ev.ycb_thresholds()
    assert len(th) == 100 and th[0] == pytest.approx(0.0005) and th[-1] == pytest.approx(0.0995)  # :18-22
    cube = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32) * 0.05
    body = ev.YCBBodyEvaluation(cube)
    # pure translation d: ADD = |d|; ADD-S = |d| too while every vertex stays closest to its own image
    add, adds = body.errors(pose(t=(0.01, 0, 0)), pose())
    assert add == pytest.approx(0.01, abs=1e-7) and adds == pytest.approx(0.01, abs=1e-7)
    # a symmetry of the cube: ADD sees the rotation, ADD-S does not
    add, adds = body.errors(pose(rot((0, 0, 1), np.pi / 2)), pose())
    assert add == pytest.approx(0.05 * 2.0, abs=1e-6) and adds == pytest.approx(0.0, abs=1e-6)
    r = body.result(pose(t=(0.0123, 0, 0)), pose())
    assert r["add_auc"] == pytest.approx(1.0 - 0.123, abs=1e-5)  # 1 - min(e / 0.1, 1) :846-847
    assert r["add_curve"].tolist() == [0.0] * 12 + [1.0] * 88    # thresholds 0.0005 … 0.0115 lie below the error
    assert body.result(pose(t=(0.5, 0, 0)), pose())["adds_auc"] == 0.0
    # the body2world pose enters inverted: identical non-trivial poses give zero error
    p = pose(rot((1, 1, 0), 0.7), (0.3, -0.2, 0.9))
    assert body.errors(p, p) == (pytest.approx(0.0, abs=1e-6), pytest.approx(0.0, abs=1e-6))


def test_ycb_reduced_vertices_and_pose_file(tmp_path):
    v = np.arange(3000, dtype=np.float32).reshape(1000, 3)
    assert ev.reduce_vertices(v, -1) is not None and len(ev.reduce_vertices(v, 0)) == 1000
    assert len(ev.reduce_vertices(v, 1000)) == 1000
    r = ev.reduce_vertices(v, 5)
    # std::mt19937{7}: 327741615, 976413892, 3349725721, 1369975286, 1882953283 (mod 1000)
    assert (r[:, 0] / 3).astype(int).tolist() == [615, 892, 721, 286, 283]
    path = tmp_path / "002_master_chef_can.txt"
    q = np.array([0.5, 0.5, 0.5, 0.5])
    with open(path, "w") as f:
        for i in range(12):
            f.write("%g %g %g %g %g %g %g\n" % (*(2 * q), 0.1 * i, 0.0, 1.0))  # unnormalised on purpose
    poses = ev.read_poses_ycb(str(path), pose_begin=2, n_frames=8, keyframes=[1, 4, 8])
    assert poses.shape == (3, 4, 4)
    assert np.allclose(poses[:, 0, 3], [0.2, 0.5, 0.9], atol=1e-6)  # lines 2 + {1, 4, 8} - 1
    assert np.allclose(poses[0, :3, :3], [[0, 0, 1], [1, 0, 0], [0, 1, 0]], atol=1e-6)  # q = (1,1,1,1)/2


def test_rbot_loop_resets_a_lost_body():
    """EvaluateRunConfiguration :174-210 on a synthetic sequence: one frame whose ground truth lies 20 cm off
    counts as lost, the body is reset to that ground truth, is lost again on the next (honest) frame because it
    now starts 20 cm off, is reset once more and tracks from there on"""
    api = util.open_oracle()
    n_frames = 6
    inputs = scenes.Inputs(1, n_frames + 1)
    inst = scenes.Instance(api, inputs)
    poses_gt = np.asarray(inputs.gt[0], np.float32).copy()
    bad = 3
    poses_gt[bad, :3, 3] += (0.2, 0.0, 0.0)
    frames, avg = ev.evaluate_rbot_sequence(inst.tracker, inst.bodies[0], poses_gt, inst.upload_frame,
                                            n_frames=n_frames)
    success = [f["tracking_success"] for f in frames]
    assert success == [1.0, 1.0, 0.0, 0.0, 1.0, 1.0]  # cycle i is judged against poses_gt[i + 1]
    assert frames[bad - 1]["translation_error"] == pytest.approx(0.2, abs=5e-3)
    assert avg["tracking_success"] == pytest.approx(4.0 / 6.0)
    assert all(f["complete_cycle"] > 0 for f in frames)


def test_ycb_loop_on_a_synthetic_sequence():
    """EvaluateRunConfiguration :333-372: Region + Depth tracking of two bodies over four keyframes, started at
    the ground truth of the first; ADD-S stays in the millimetre range and the AUC near 1"""
    api = util.open_oracle()
    keyframes = [1, 2, 4, 5]  # 1-based image numbers like the dataset's keyframe lists
    inputs = scenes.Inputs(2, 6, with_depth=True)
    inst = scenes.Instance(api, inputs, use_region=True, use_depth=True)
    names = ["body0", "body1"]
    bodies = dict(zip(names, inst.bodies))
    evaluations = {n: ev.YCBBodyEvaluation(inputs.vertices[i], 200) for i, n in enumerate(names)}
    gt = {n: np.asarray([inputs.gt[i][k - 1] for k in keyframes], np.float32) for i, n in enumerate(names)}
    results, average = ev.evaluate_ycb_sequence(inst.tracker, bodies, evaluations, gt, keyframes,
                                                lambda k: inst.upload_frame(k - 1))
    for n in names:
        assert len(results[n]) == len(keyframes)
        assert max(r["adds_error"] for r in results[n]) < 5e-3
        assert average[n]["adds_auc"] > 0.95 and average[n]["add_auc"] > 0.9
        assert average[n]["adds_curve"][-1] == 1.0 and average[n]["adds_curve"].shape == (100,)


def write_rbot_dataset(tmp_path, n_frames=5):
    """a two-body, one-sequence dataset of the synthetic scenes in the RBOT layout, with the region models the
    device would generate replaced by the synthetic bodies' (the oracle cannot generate); returns
    (dataset, external, body names, model parameters)"""
    from PIL import Image
    cfg = util.pkg.config
    dataset, external = tmp_path / "RBOT_dataset", tmp_path / "external"
    scenes_ = [util.syn.Scene(i, intr=dict(zip(("fu", "fv", "ppu", "ppv", "width", "height"), ev.RBOT_INTRINSICS)))
               for i in range(2)]
    trajectory = [scenes_[0].pose.copy()]
    for _ in range(n_frames):
        scenes_[0].step_pose()
        trajectory.append(scenes_[0].pose.copy())
    with open(dataset.parent / "poses.tmp", "w") as f:
        f.write("header\n")
        for p in trajectory:
            f.write("\t".join("%.9g" % v for v in list(p[:3, :3].reshape(-1)) + list(p[:3, 3] * 1000.0)) + "\n")
    os.makedirs(dataset)
    os.replace(dataset.parent / "poses.tmp", dataset / "poses_first.txt")
    model_parameters = dict(ev.RBOT_MODEL_PARAMETERS, n_divides=2)
    octahedron = [(60, 0, 0), (-60, 0, 0), (0, 50, 0), (0, -50, 0), (0, 0, 40), (0, 0, -40)]
    faces = [(1, 3, 5), (3, 2, 5), (2, 4, 5), (4, 1, 5), (3, 1, 6), (2, 3, 6), (4, 2, 6), (1, 4, 6)]
    names = ["ape", "cat"]
    for name, scene in zip(names, scenes_):
        os.makedirs(dataset / name / "frames")
        with open(dataset / name / (name + ".obj"), "w") as f:
            f.writelines("v %d %d %d\n" % v for v in octahedron)
            f.writelines("f %d %d %d\n" % t for t in faces)
        for k, p in enumerate(trajectory):
            Image.fromarray(np.ascontiguousarray(scene.render(p)[:, :, ::-1])).save(
                dataset / name / "frames" / ("a_regular%04d.png" % k))
        # the model the device would generate from the mesh, here the synthetic body's (the oracle cannot generate)
        points, orientations, lengths = util.syn.make_region_model(scene.body, n_divides=2, n_points=200)
        vertices, _ = cfg.load_obj(str(dataset / name / (name + ".obj")), 0.001)
        data = cfg.BodyData(str(dataset / name / (name + ".obj")), 0.001, True, False,
                            cfg.maximum_body_diameter(vertices), np.eye(4))
        cfg.write_model_bin(str(external / "models" / (name + "_model.bin")), True, model_parameters, data, points,
                            orientations, lengths)
    return dataset, external, names, model_parameters


def test_rbot_dataset_driver_on_a_synthetic_dataset_in_the_rbot_layout(tmp_path):
    """evaluate_rbot_dataset (examples/evaluate_rbot_dataset.cpp): bodies as <dataset>/<body>/<body>.obj in mm,
    frames <dataset>/<body>/frames/<sequence>NNNN.png, one poses_first.txt for all bodies, models under
    <external>/models/ — a two-body, one-sequence, five-frame dataset of the synthetic scenes, tracked by the oracle"""
    n_frames = 5
    dataset, external, names, model_parameters = write_rbot_dataset(tmp_path, n_frames)
    titles = []
    results, overall = ev.evaluate_rbot_dataset(util.open_oracle, str(dataset), str(external), names, ["a_regular"],
                                                n_frames=n_frames, model_parameters=model_parameters,
                                                report=lambda title, r: titles.append(title))
    assert titles == ["a_regular_ape", "a_regular_cat"] and set(results) == {("a_regular", "ape"), ("a_regular", "cat")}
    for r in results.values():
        assert r["tracking_success"] == 1.0 and r["translation_error"] < 5e-3 and r["rotation_error"] < np.deg2rad(2)
    assert overall["tracking_success"] == 1.0
    # two processes sharing the runs: together they cover the same (sequence, body) pairs with the same results
    merged = {}
    for rank in range(2):
        part, _ = ev.evaluate_rbot_dataset(util.open_oracle, str(dataset), str(external), names, ["a_regular"],
                                           n_frames=n_frames, model_parameters=model_parameters, shard=(rank, 2))
        assert len(part) == 1
        merged.update(part)
    assert set(merged) == set(results)
    for key in results:
        assert merged[key]["translation_error"] == results[key]["translation_error"]
    # a model made with other parameters is not accepted, and the oracle context cannot replace it
    with pytest.raises(util.pkg.M3TError):
        ev.evaluate_rbot_dataset(util.open_oracle, str(dataset), str(external), names[:1], ["a_regular"],
                                 n_frames=n_frames)


def _pose_line(p):
    """'qw qx qy qz tx ty tz' of a 4x4 pose (w >= 0)"""
    m = np.asarray(p, np.float64)
    w = np.sqrt(max(0.0, 1.0 + m[0, 0] + m[1, 1] + m[2, 2])) / 2.0
    x, y, z = (m[2, 1] - m[1, 2]) / (4 * w), (m[0, 2] - m[2, 0]) / (4 * w), (m[1, 0] - m[0, 1]) / (4 * w)
    return " ".join("%.9g" % v for v in (w, x, y, z, m[0, 3], m[1, 3], m[2, 3]))


def write_ycb_dataset(tmp_path):
    """two sequences with one synthetic body each in the YCB-Video layout, ground truth in both forms, models
    pre-written; returns (dataset, external, body names, model parameters)"""
    from PIL import Image
    cfg = util.pkg.config
    dataset, external = tmp_path / "YCB-Video", tmp_path / "external"
    intr = dict(zip(("fu", "fv", "ppu", "ppv", "width", "height"), ev.YCB_INTRINSICS))
    names = ["002_master_chef_can", "003_cracker_box"]
    keyframes = {"0000": [1, 2, 4, 5], "0001": [1, 3]}
    n_frames = {"0000": 5, "0001": 3}
    model_parameters = dict(ev.YCB_MODEL_PARAMETERS, n_divides=2, n_points=200)
    octahedron = [(0.06, 0, 0), (-0.06, 0, 0), (0, 0.05, 0), (0, -0.05, 0), (0, 0, 0.04), (0, 0, -0.04)]
    faces = [(1, 3, 5), (3, 2, 5), (2, 4, 5), (4, 1, 5), (3, 1, 6), (2, 3, 6), (4, 2, 6), (1, 4, 6)]
    os.makedirs(dataset / "image_sets")
    os.makedirs(dataset / "poses")
    os.makedirs(external / "poses" / "ground_truth")
    with open(dataset / "image_sets" / "keyframe.txt", "w") as f:
        f.writelines("%s/%06d\n" % (s, k) for s in keyframes for k in keyframes[s])
    for index, (sequence, name) in enumerate(zip(("0000", "0001"), names)):
        scene = util.syn.Scene(index, intr=intr, with_depth=True, depth_scale=1e-4)
        os.makedirs(dataset / "data" / sequence)
        os.makedirs(dataset / "models" / name)
        with open(dataset / "models" / name / "textured.obj", "w") as f:
            f.writelines("v %g %g %g\n" % v for v in octahedron)
            f.writelines("f %d %d %d\n" % t for t in faces)
        poses = []
        for k in range(1, n_frames[sequence] + 1):
            if k > 1:
                scene.step_pose()
            poses.append(scene.pose.copy())
            color, depth = scene.render()
            Image.fromarray(np.ascontiguousarray(color[:, :, ::-1])).save(dataset / "data" / sequence / ("%06d-color.png" % k))
            Image.fromarray(depth).save(dataset / "data" / sequence / ("%06d-depth.png" % k))
            (dataset / "data" / sequence / ("%06d-box.txt" % k)).write_text("%s 10.0 20.0 110.0 120.0\n" % name)
        (dataset / "poses" / (name + ".txt")).write_text("".join(_pose_line(p) + "\n" for p in poses))
        (external / "poses" / "ground_truth" / ("%s_%s.txt" % (sequence, name))).write_text(
            "".join(_pose_line(poses[k - 1]) + "\n" for k in keyframes[sequence]))
        vertices, _ = cfg.load_obj(str(dataset / "models" / name / "textured.obj"))
        data = cfg.BodyData(str(dataset / "models" / name / "textured.obj"), 1.0, True, True,
                            cfg.maximum_body_diameter(vertices), np.eye(4))
        rp, ro, rl = util.syn.make_region_model(scene.body, n_divides=2, n_points=200)
        cfg.write_model_bin(str(external / "models" / (name + "_region_model.bin")), True, model_parameters, data, rp, ro, rl)
        dp, do, da = util.syn.make_depth_model(scene.body, n_divides=2, n_points=200)
        cfg.write_model_bin(str(external / "models" / (name + "_depth_model.bin")), False, model_parameters, data, dp, do, da)
    return dataset, external, names, model_parameters


def test_ycb_dataset_driver_on_a_synthetic_dataset_in_the_ycb_layout(tmp_path):
    """evaluate_ycb_dataset (examples/evaluate_ycb_dataset.cpp): models/<body>/textured.obj, data/<sequence>/
    NNNNNN-{color,depth}.png + NNNNNN-box.txt, image_sets/keyframe.txt, ground truth per keyframe under
    external/poses/ground_truth/ or per frame in the dataset's poses/<body>.txt — two sequences with one synthetic
    body each, Region + Depth with measured occlusions on the oracle"""
    dataset, external, names, model_parameters = write_ycb_dataset(tmp_path)
    assert ev.ycb_keyframes(str(dataset), "0000") == [1, 2, 4, 5] and ev.ycb_n_frames(str(dataset), "0001") == 3
    assert ev.ycb_sequence_bodies(str(dataset), "0001") == [names[1]]
    titles = []
    outcomes = []
    for matlab in (True, False):
        results, overall = ev.evaluate_ycb_dataset(util.open_oracle, str(dataset), str(external), [0, 1], names,
                                                   use_matlab_gt_poses=matlab, n_vertices_evaluation=4,
                                                   model_parameters=model_parameters,
                                                   report=lambda title, r: titles.append(title))
        assert set(results) == {("0000", names[0]), ("0001", names[1])}  # only the bodies a sequence contains
        # the octahedron stands in for the tracked shape: ADD-S over its vertices still measures the pose error
        assert overall["adds_auc"] > 0.9 and overall["add_auc"] > 0.85
        outcomes.append((overall["add_auc"], overall["adds_auc"]))
    assert titles[:2] == ["0000: " + names[0], "0001: " + names[1]]
    assert outcomes[0] == pytest.approx(outcomes[1], abs=1e-4)  # both ground-truth sources describe the same poses
