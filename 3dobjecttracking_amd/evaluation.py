"""Evaluator front-ends of the reference's dataset benchmarks, over the batched tracker.

RBOT (M3T/examples/rbot_evaluator.cpp): ground-truth pose file reader, the 5 cm / 5 degree success criterion,
the frame loop with reset-on-loss.  YCB-Video (M3T/examples/ycb_evaluator.cpp): ground-truth reader, ADD /
ADD-S per frame, the tracking-loss curves and the area under curve, reduced evaluation vertices.  The
datasets themselves are external downloads; the loops take a frame source, so that synthetic sequences and
the real datasets run through the same code.
"""
import time

import numpy as np

F = np.float32


# ---------------------------------------------------------------------------------------------------------
# RBOT
# ---------------------------------------------------------------------------------------------------------
def read_poses_rbot(path, n_frames=1000):
    """RBOTEvaluator::ReadPosesRBOTDataset (rbot_evaluator.cpp:558-585): one header line, then per frame
    nine rotation entries (row-major) and a translation in millimetres, tab separated; n_frames + 1 poses."""
    poses = np.zeros((n_frames + 1, 4, 4), F)
    with open(path) as f:
        f.readline()
        for i in range(n_frames + 1):
            t = f.readline().split("\t")
            if len(t) < 12:
                raise ValueError("Could not read pose %d from %s" % (i, path))
            v = [F(x) for x in t[:12]]
            poses[i, :3, :3] = np.asarray(v[:9], F).reshape(3, 3)
            poses[i, :3, 3] = np.asarray(v[9:12], F) * F(0.001)
            poses[i, 3, 3] = 1.0
    return poses


def rbot_pose_result(body2world_pose, body2world_pose_gt, translation_error_threshold=0.05,
                     rotation_error_threshold=5.0 * np.pi / 180.0):
    """RBOTEvaluator::CalculatePoseResults (rbot_evaluator.cpp:416-433): (translation error [m], rotation error
    [rad], tracking success 0/1).  Thresholds: rbot_evaluator.h:192-193."""
    p, g = np.asarray(body2world_pose, F), np.asarray(body2world_pose_gt, F)
    t_err = float(np.linalg.norm((p[:3, 3] - g[:3, 3]).astype(F)))
    c = (np.trace((p[:3, :3].T @ g[:3, :3]).astype(F)) - F(1.0)) / F(2.0)
    r_err = float(np.arccos(c))  # NaN for |c| > 1 exactly like acos(); the comparisons below are then false
    lost = t_err > translation_error_threshold or r_err > rotation_error_threshold
    return t_err, r_err, 0.0 if lost else 1.0


def evaluate_rbot_sequence(tracker, body, poses_gt, load_image, n_frames=None, reset_renderers=()):
    """RBOTEvaluator::EvaluateRunConfiguration (rbot_evaluator.cpp:174-210) for the main body: start on image 0
    at its ground truth, then cycle i tracks image i + 1 (the loader camera advances once per UpdateCameras),
    compares with the ground truth of image i + 1 and resets the body to it (ResetBody :334-342) whenever
    tracking is lost.  `load_image(k)` makes image k the cameras' current image.
    Returns the per-frame results and their average (CalculateAverageResult :435-470)."""
    n_frames = len(poses_gt) - 1 if n_frames is None else n_frames

    def reset(i):
        body.set_body2world_pose(poses_gt[i])
        for r in reset_renderers:
            r.StartRendering()
        tracker.StartModalities(0)

    load_image(0)
    reset(0)
    frames = []
    for i in range(n_frames):
        load_image(i + 1)
        t0 = time.perf_counter()
        ok = tracker.ExecuteTrackingStep(i) and tracker.Sync()
        dt = (time.perf_counter() - t0) * 1e6
        if not ok:
            raise RuntimeError("tracking step %d failed" % i)
        t_err, r_err, success = rbot_pose_result(body.body2world_pose(), poses_gt[i + 1])
        frames.append(dict(frame_index=i, translation_error=t_err, rotation_error=r_err,
                           tracking_success=success, complete_cycle=dt))
        if success == 0.0:
            reset(i + 1)
    avg = {k: float(np.mean([f[k] for f in frames])) for k in
           ("translation_error", "rotation_error", "tracking_success", "complete_cycle")}
    return frames, avg


# ---------------------------------------------------------------------------------------------------------
# YCB-Video
# ---------------------------------------------------------------------------------------------------------
K_N_CURVE_VALUES = 100   # ycb_evaluator.h:45
K_THRESHOLD_MAX = 0.1    # ycb_evaluator.h:46


def ycb_thresholds():
    """ycb_evaluator.cpp:18-22"""
    step = F(K_THRESHOLD_MAX) / F(K_N_CURVE_VALUES)
    return (step * (F(0.5) + np.arange(K_N_CURVE_VALUES, dtype=F))).astype(F)


def _quaternion_pose(w, x, y, z, tx, ty, tz):
    q = np.asarray([w, x, y, z], F)
    q = q / F(np.sqrt((q * q).sum(dtype=F)))
    w, x, y, z = [float(v) for v in q]
    pose = np.eye(4, dtype=F)
    pose[:3, :3] = np.asarray([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                               [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                               [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], F)
    pose[:3, 3] = (tx, ty, tz)
    return pose


def read_poses_ycb(path, pose_begin, n_frames, keyframes):
    """YCBEvaluator::LoadGTPoses (ycb_evaluator.cpp:850-901): the per-body pose file holds one line
    'qw qx qy qz tx ty tz' per frame of every sequence; skip `pose_begin` lines, then keep the lines whose
    1-based frame index is a keyframe.  Quaternions are normalised, pose = translation * rotation."""
    keyframes = list(keyframes)
    poses = []
    with open(path) as f:
        for _ in range(pose_begin):
            f.readline()
        k = 0
        for idx in range(1, n_frames + 1):
            if k >= len(keyframes):
                break
            line = f.readline()
            if idx == keyframes[k]:
                v = [float(x) for x in line.split(" ")[:7]]
                poses.append(_quaternion_pose(*v))
                k += 1
    return np.asarray(poses, F)


def reduce_vertices(vertices, n_vertices_evaluation):
    """YCBEvaluator::GenderateReducedVertices (ycb_evaluator.cpp:1222-1248): all vertices, or
    n_vertices_evaluation draws `mt19937{7}() % n` with repetition"""
    vertices = np.asarray(vertices, F)
    n = len(vertices)
    if n_vertices_evaluation <= 0 or n_vertices_evaluation >= n:
        return vertices
    raw = np.random.RandomState(7)._bit_generator.random_raw(n_vertices_evaluation)  # == std::mt19937{7}
    return vertices[(raw % n).astype(np.int64)]


class YCBBodyEvaluation:
    """the per-body data of YCBEvaluator::CalculatePoseResults (ycb_evaluator.cpp:803-848): reduced vertices
    and a nearest-neighbour index over them (nanoflann there, scipy's k-d tree here: both exact)"""

    def __init__(self, vertices, n_vertices_evaluation=-1):
        from scipy.spatial import cKDTree
        self.vertices = reduce_vertices(vertices, n_vertices_evaluation)
        self.tree = cKDTree(self.vertices.astype(np.float64))
        self.thresholds = ycb_thresholds()

    def errors(self, body2world_pose, gt_body2world_pose):
        """(ADD, ADD-S) in metres: delta = body2world^-1 * gt; mean |v - delta v| and mean nearest-vertex
        distance of delta v"""
        p = np.asarray(body2world_pose, np.float64)
        g = np.asarray(gt_body2world_pose, np.float64)
        pi = np.eye(4)
        pi[:3, :3] = p[:3, :3].T
        pi[:3, 3] = -p[:3, :3].T @ p[:3, 3]
        delta = (pi @ g).astype(F)
        v = (self.vertices @ delta[:3, :3].T + delta[:3, 3]).astype(F)
        add = float(np.sqrt(((self.vertices - v) ** 2).sum(axis=1, dtype=F)).mean(dtype=F))
        dist, _ = self.tree.query(v.astype(np.float64), k=1)
        adds = float(dist.astype(F).mean(dtype=F))
        return add, adds

    def result(self, body2world_pose, gt_body2world_pose):
        add, adds = self.errors(body2world_pose, gt_body2world_pose)
        out = dict(add_error=add, adds_error=adds)
        for key, err in (("add", add), ("adds", adds)):
            curve = np.ones(K_N_CURVE_VALUES, F)
            for i in range(K_N_CURVE_VALUES):  # the curve is 0 below the error, 1 from the first threshold above
                if err < self.thresholds[i]:
                    break
                curve[i] = 0.0
            out[key + "_curve"] = curve
            out[key + "_auc"] = float(F(1.0) - min(F(err) / F(K_THRESHOLD_MAX), F(1.0)))
        return out


def evaluate_ycb_sequence(tracker, bodies, evaluations, gt_body2world_poses, keyframes, update_cameras):
    """YCBEvaluator::EvaluateRunConfiguration without refinement (ycb_evaluator.cpp:333-372): bodies start at
    the ground truth of the first keyframe, StartModalities once, then one tracking step per keyframe and the
    ADD / ADD-S results of every evaluated body.  bodies / evaluations / gt poses are dicts by body name."""
    for name, body in bodies.items():
        body.set_body2world_pose(gt_body2world_poses[name][0])
    update_cameras(keyframes[0])
    if not tracker.StartModalities(0):
        raise RuntimeError("StartModalities failed")
    results = {name: [] for name in evaluations}
    for i, frame in enumerate(keyframes):
        update_cameras(frame)
        t0 = time.perf_counter()
        if not (tracker.ExecuteTrackingStep(i) and tracker.Sync()):
            raise RuntimeError("tracking step %d failed" % i)
        dt = (time.perf_counter() - t0) * 1e6
        for name, ev in evaluations.items():
            r = ev.result(bodies[name].body2world_pose(), gt_body2world_poses[name][i])
            r.update(frame_index=i, complete_cycle=dt)
            results[name].append(r)
    average = {}
    for name, rs in results.items():
        average[name] = dict(add_auc=float(np.mean([r["add_auc"] for r in rs])),
                             adds_auc=float(np.mean([r["adds_auc"] for r in rs])),
                             add_curve=np.mean([r["add_curve"] for r in rs], axis=0),
                             adds_curve=np.mean([r["adds_curve"] for r in rs], axis=0),
                             complete_cycle=float(np.mean([r["complete_cycle"] for r in rs])))
    return results, average


# ---------------------------------------------------------------------------------------------------------
# RBOT dataset driver (examples/evaluate_rbot_dataset.cpp + rbot_evaluator.cpp)
# ---------------------------------------------------------------------------------------------------------
RBOT_INTRINSICS = (650.048, 647.183, 324.328 - 0.5, 257.323 - 0.5, 640, 512)  # rbot_evaluator.h:40-41
RBOT_BODY_NAMES = ("ape", "bakingsoda", "benchviseblue", "broccolisoup", "cam", "can", "cat", "clown", "cube", "driller",
                   "duck", "eggbox", "glue", "iron", "koalacandy", "lamp", "phone", "squirrel")
RBOT_SEQUENCE_NAMES = ("a_regular", "b_dynamiclight", "c_noisy", "d_occlusion")
RBOT_REGION_PARAMETERS = dict(  # evaluate_rbot_dataset.cpp:25-44, rbot_evaluator.cpp:267
    n_lines_max=200, use_adaptive_coverage=0, min_continuous_distance=3.0, function_length=8, distribution_length=12,
    function_amplitude=0.36, function_slope=0.0, learning_rate=1.3, scales=[5, 2, 2, 1],
    standard_deviations=[20.0, 7.0, 3.0, 1.5], n_histogram_bins=32, learning_rate_f=0.2, learning_rate_b=0.2,
    unconsidered_line_length=0.5, max_considered_line_length=20.0, n_unoccluded_iterations=0)
RBOT_MODEL_PARAMETERS = dict(sphere_radius=0.8, n_divides=4, n_points=200, max_radius_depth_offset=0.01,
                             stride_depth_offset=0.002, use_random_seed=False, image_size=2000)  # :548-551


def evaluate_rbot_dataset(open_context, dataset_directory, external_directory, body_names=RBOT_BODY_NAMES,
                          sequence_names=RBOT_SEQUENCE_NAMES, n_frames=1000, region_parameters=None,
                          model_parameters=None, tikhonov_parameter_rotation=1000.0,
                          tikhonov_parameter_translation=30000.0, n_corr_iterations=7, n_update_iterations=2,
                          report=None, shard=(0, 1)):
    """RBOTEvaluator::SetUp + Evaluate for the region modality on the un-modelled sequences: for every (sequence,
    body) a tracker on `dataset/<body>/frames/<sequence>NNNN.png`, started at `dataset/poses_first.txt`, reset on
    loss, scored with the 5 cm / 5 degree criterion.  Bodies are `dataset/<body>/<body>.obj` in millimetres
    (LoadSingleBody :527-535), their region models `external/models/<body>_model.bin` — generated on the device
    when missing or made with other parameters (GenerateSingleModel :546-556).  `open_context()` returns a fresh
    device context per run (one tracker per context).  Returns {(sequence, body): average result} and the overall
    average (CalculateAverageResult); `report`, if given, is called with each run's title and result.
    shard = (rank, world): this process takes every world-th run (the reference spreads the runs over OpenMP
    threads, rbot_evaluator.cpp:139-156; here one process per GPU takes its share and the caller merges the
    dictionaries)."""
    import os

    from . import config as cfg
    from . import generator, host
    poses_first = read_poses_rbot(os.path.join(dataset_directory, "poses_first.txt"), n_frames)
    region_parameters = dict(RBOT_REGION_PARAMETERS, **(region_parameters or {}))
    model_parameters = dict(RBOT_MODEL_PARAMETERS, **(model_parameters or {}))
    results = {}
    runs = [(sequence, name) for sequence in sequence_names for name in body_names]
    for sequence, name in runs[shard[0]::shard[1]]:
        api = open_context()
        body = generator.Body(api, name, os.path.join(dataset_directory, name, name + ".obj"), 0.001, True, False,
                              np.eye(4, dtype=F))
        model_path = os.path.join(external_directory, "models", name + "_model.bin")
        if cfg.model_bin_matches(model_path, True, model_parameters, body.body_data()):
            model = host.RegionModel(api, path=model_path)
        else:
            generation = {k: v for k, v in model_parameters.items() if k != "use_random_seed"}
            model = host.RegionModel.generate(api, body, **generation)
            cfg.write_model_bin(model_path, True, model_parameters, body.body_data(), *model.views())
        camera = generator.LoaderColorCamera(api, os.path.join(dataset_directory, name, "frames"), RBOT_INTRINSICS,
                                             sequence, 0, 4)
        modality = host.RegionModality(api, body, camera, model, **region_parameters)
        host.Optimizer(api, body=body, modalities=[modality],
                       tikhonov_parameter_rotation=tikhonov_parameter_rotation,
                       tikhonov_parameter_translation=tikhonov_parameter_translation)
        tracker = host.Tracker(api, n_corr_iterations, n_update_iterations)

        def load_image(k, camera=camera):
            camera.set_load_index(k)
            if not camera.UpdateImage():
                raise RuntimeError("Could not read image from %s" % camera.image_path())

        _, average = evaluate_rbot_sequence(tracker, body, poses_first, load_image, n_frames)
        results[(sequence, name)] = average
        if report is not None:
            report(sequence + "_" + name, average)
    keys = ("translation_error", "rotation_error", "tracking_success", "complete_cycle")
    overall = {k: float(np.mean([r[k] for r in results.values()])) for k in keys}
    return results, overall


# ---------------------------------------------------------------------------------------------------------
# YCB-Video dataset driver (examples/evaluate_ycb_dataset.cpp + ycb_evaluator.cpp), without refinement, without
# modelled occlusions, single-region models
# ---------------------------------------------------------------------------------------------------------
YCB_INTRINSICS = (1066.778, 1067.487, 312.9869, 241.3109, 640, 480)  # ycb_evaluator.h:47-48
YCB_REGION_PARAMETERS = dict(  # evaluate_ycb_dataset.cpp:46-65
    n_lines_max=200, use_adaptive_coverage=0, min_continuous_distance=3.0, function_length=8, distribution_length=12,
    function_amplitude=0.43, function_slope=0.5, learning_rate=1.3, scales=[7, 4, 2],
    standard_deviations=[25.0, 15.0, 10.0], n_histogram_bins=16, learning_rate_f=0.2, learning_rate_b=0.2,
    unconsidered_line_length=0.5, max_considered_line_length=20.0, measured_depth_offset_radius=0.01,
    measured_occlusion_radius=0.01, measured_occlusion_threshold=0.03, n_unoccluded_iterations=0)
YCB_DEPTH_PARAMETERS = dict(  # evaluate_ycb_dataset.cpp:66-76
    n_points_max=200, use_adaptive_coverage=0, use_depth_scaling=0, stride_length=0.005,
    considered_distances=[0.07, 0.05, 0.04], standard_deviations=[0.05, 0.03, 0.02],
    measured_depth_offset_radius=0.01, measured_occlusion_radius=0.01, measured_occlusion_threshold=0.03,
    n_unoccluded_iterations=0)
YCB_MODEL_PARAMETERS = dict(sphere_radius=0.8, n_divides=4, n_points=500, max_radius_depth_offset=0.05,
                            stride_depth_offset=0.002, use_random_seed=False, image_size=2000)  # :1131-1146


def ycb_sequence_name(sequence_id):
    return "%04d" % sequence_id  # SequenceIDToName :1312-1315


def ycb_keyframes(dataset_directory, sequence_name):
    """LoadKeyframes :1150-1183: image_sets/keyframe.txt holds lines '<sequence>/<frame>'"""
    import os
    frames = []
    with open(os.path.join(dataset_directory, "image_sets", "keyframe.txt")) as f:
        for line in f:
            sequence, _, frame = line.strip().partition("/")
            if sequence == sequence_name and frame:
                frames.append(int(frame))
    return frames


def ycb_sequence_bodies(dataset_directory, sequence_name):
    """SequenceBodyNames / BodyExistsInSequence :1262-1300: the first word of every line of 000001-box.txt"""
    import os
    with open(os.path.join(dataset_directory, "data", sequence_name, "000001-box.txt")) as f:
        return [line.split(" ")[0] for line in f if line.strip()]


def ycb_n_frames(dataset_directory, sequence_name):
    """NFramesInSequence :1302-1310"""
    import os
    i = 1
    while os.path.exists(os.path.join(dataset_directory, "data", sequence_name, "%06d-box.txt" % i)):
        i += 1
    return i - 1


def read_matlab_poses_ycb(path):
    """LoadMatlabGTPoses :903-944: one 'qw qx qy qz tx ty tz' line per keyframe"""
    with open(path) as f:
        return np.asarray([_quaternion_pose(*[float(x) for x in line.split(" ")[:7]]) for line in f if line.strip()], F)


def evaluate_ycb_dataset(open_context, dataset_directory, external_directory, sequence_ids, body_names,
                         use_matlab_gt_poses=True, n_vertices_evaluation=1000, region_parameters=None,
                         depth_parameters=None, model_parameters=None, tikhonov_parameter_rotation=1000.0,
                         tikhonov_parameter_translation=30000.0, n_corr_iterations=4, n_update_iterations=2,
                         report=None, shard=(0, 1)):
    """YCBEvaluator::SetUp + Evaluate with the region and the depth modality, measured occlusions, one run per
    (sequence, body present in it) (CreateRunConfigurations :1006-1022): bodies `dataset/models/<body>/textured.obj`
    in metres, frames `dataset/data/<sequence>/NNNNNN-{color,depth}.png` (depth scale 1e-4), keyframes from
    `dataset/image_sets/keyframe.txt`, ground truth from `external/poses/ground_truth/<sequence>_<body>.txt` (or the
    dataset's own `poses/<body>.txt`), models under `external/models/`.  Returns {(sequence, body): average} and
    the averages over all frames of all runs (CalculateAverageResult).  shard = (rank, world): every world-th run
    (see evaluate_rbot_dataset); the overall averages then cover this process's runs."""
    import os

    from . import config as cfg
    from . import generator, host
    region_parameters = dict(YCB_REGION_PARAMETERS, **(region_parameters or {}))
    depth_parameters = dict(YCB_DEPTH_PARAMETERS, **(depth_parameters or {}))
    model_parameters = dict(YCB_MODEL_PARAMETERS, **(model_parameters or {}))
    generation = {k: v for k, v in model_parameters.items() if k != "use_random_seed"}
    sequence_names = [ycb_sequence_name(i) for i in sequence_ids]
    n_frames = {ycb_sequence_name(i): ycb_n_frames(dataset_directory, ycb_sequence_name(i))
                for i in range(max(sequence_ids) + 1)
                if os.path.isdir(os.path.join(dataset_directory, "data", ycb_sequence_name(i)))}
    results, frame_results = {}, []
    runs = [(sequence, name) for sequence in sequence_names
            for name in body_names if name in ycb_sequence_bodies(dataset_directory, sequence)]
    for sequence, name in runs[shard[0]::shard[1]]:
        keyframes = ycb_keyframes(dataset_directory, sequence)
        api = open_context()
        body = generator.Body(api, name, os.path.join(dataset_directory, "models", name, "textured.obj"), 1.0, True,
                              True, np.eye(4, dtype=F))
        models = []
        for region, klass, suffix in ((True, host.RegionModel, "_region_model.bin"),
                                      (False, host.DepthModel, "_depth_model.bin")):
            path = os.path.join(external_directory, "models", name + suffix)
            if cfg.model_bin_matches(path, region, model_parameters, body.body_data()):
                models.append(klass(api, path=path))
            else:
                models.append(klass.generate(api, body, **generation))
                cfg.write_model_bin(path, region, model_parameters, body.body_data(), *models[-1].views())
        directory = os.path.join(dataset_directory, "data", sequence)
        color = generator.LoaderColorCamera(api, directory, YCB_INTRINSICS, "", 1, 6, "-color")
        depth = generator.LoaderDepthCamera(api, directory, YCB_INTRINSICS, 0.0001, "", 1, 6, "-depth")
        region_modality = host.RegionModality(api, body, color, models[0], depth_camera=depth, measure_occlusions=1,
                                              **region_parameters)
        depth_modality = host.DepthModality(api, body, depth, models[1], measure_occlusions=1, **depth_parameters)
        host.Optimizer(api, body=body, modalities=[region_modality, depth_modality],
                       tikhonov_parameter_rotation=tikhonov_parameter_rotation,
                       tikhonov_parameter_translation=tikhonov_parameter_translation)
        tracker = host.Tracker(api, n_corr_iterations, n_update_iterations)
        if use_matlab_gt_poses:
            gt = read_matlab_poses_ycb(os.path.join(external_directory, "poses", "ground_truth",
                                                    sequence + "_" + name + ".txt"))
        else:  # the dataset's pose file: the body's frames of all earlier sequences come first (LoadPoseBegin)
            begin = sum(n_frames[s] for s in sorted(n_frames) if s < sequence and
                        name in ycb_sequence_bodies(dataset_directory, s))
            gt = read_poses_ycb(os.path.join(dataset_directory, "poses", name + ".txt"), begin, n_frames[sequence],
                                keyframes)
        if len(gt) < len(keyframes):
            raise ValueError("ground truth of %s in sequence %s has %d poses for %d keyframes" %
                             (name, sequence, len(gt), len(keyframes)))

        def update_cameras(frame, color=color, depth=depth):
            for camera in (color, depth):
                camera.set_load_index(frame)
                if not camera.UpdateImage():
                    raise RuntimeError("Could not read image from %s" % camera.image_path())

        evaluation = YCBBodyEvaluation(body.vertices, n_vertices_evaluation)
        per_frame, average = evaluate_ycb_sequence(tracker, {name: body}, {name: evaluation}, {name: gt}, keyframes,
                                                   update_cameras)
        results[(sequence, name)] = average[name]
        frame_results += per_frame[name]
        if report is not None:
            report(sequence + ": " + name, average[name])
    overall = dict(add_auc=float(np.mean([r["add_auc"] for r in frame_results])),
                   adds_auc=float(np.mean([r["adds_auc"] for r in frame_results])),
                   complete_cycle=float(np.mean([r["complete_cycle"] for r in frame_results])))
    return results, overall
