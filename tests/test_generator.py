"""Configuration front-end: OpenCV-YAML metafiles, loader cameras, model files and GenerateConfiguredTracker
(M3T/include/m3t/generator.h) checked on the reference's own configuration of
TrackerTest.OptimizePoseMatrixGeneratorSetUp (test/tracker_test.cpp:182-195, data/tracker_test/tracker_config.yaml)
and its golden pose, with the CPU oracle as the device context."""
import os
import shutil
import sys

import numpy as np
import pytest

import util

sys.path.insert(0, os.path.join(util.ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
import gl_model  # noqa: E402
import golden_scene as gs  # noqa: E402

cfg = util.pkg.config
generator = util.pkg.generator


def reference_tree(tmp_path):
    """the reference's data directory layout with the fixture files (model_path '../../temp/…' of the model
    metafiles then lands in tmp_path/temp)"""
    root = tmp_path / "data"
    for sub in ("tracker_test", "_body", "_sequence"):
        shutil.copytree(os.path.join(util.GOLDEN, sub), root / sub)
    return root


def write_fixture_models(root):
    """the 2 x 5 views next to the fixture pose (tests/golden/triangle_views.npz) as model files with the
    header the generator expects for data/_body/triangle.yaml"""
    v = gs.views()
    vertices, _ = cfg.load_obj(str(root / "_body" / "triangle.obj"))
    g2b = np.asarray(gs.mtv.GEOMETRY2BODY, np.float32)
    moved = vertices @ g2b[:3, :3].T + g2b[:3, 3]
    bd = cfg.BodyData(str(root / "_body" / "triangle.obj"), 1.0, True, True, cfg.maximum_body_diameter(moved), g2b)
    params = dict(generator._MODEL_DEFAULTS)
    temp = root.parent / "temp"
    cfg.write_model_bin(str(temp / "triangle_region_model.bin"), True, params, bd, v["region_points"],
                        v["region_orientations"], v["region_contour_lengths"])
    cfg.write_model_bin(str(temp / "triangle_depth_model.bin"), False, params, bd, v["depth_points"],
                        v["depth_orientations"], v["depth_surface_areas"])


def test_yaml_metafiles():
    d = cfg.read_yaml(os.path.join(util.GOLDEN, "_sequence", "depth_camera.yaml"))
    assert d["intrinsics"]["width"] == 848 and d["depth_scale"] == 0.001 and d["load_index"] == 200
    assert np.allclose(d["camera2world_pose"], util.DEPTH_CAMERA2WORLD)
    b = cfg.read_yaml(os.path.join(util.GOLDEN, "_body", "triangle.yaml"))
    assert b["geometry_path"] == "triangle.obj" and b["body_id"] == 150 and b["region_id"] == 150
    assert np.array_equal(cfg.pose(b["geometry2body_pose"]), np.asarray(gs.mtv.GEOMETRY2BODY, np.float32))
    t = cfg.read_yaml(os.path.join(util.GOLDEN, "tracker_test", "tracker_config.yaml"))
    assert t["RegionModality"][0]["measure_occlusions"] == {"depth_camera": "depth_camera"}
    assert t["Link"][0]["modalities"] == ["triangle_region_modality", "triangle_depth_modality"]
    with pytest.raises(ValueError):
        cfg.required(b, ("geometry_path", "no_such_key"), "body", "triangle.yaml")


def test_obj_loader_matches_fixture_meshes():
    for name in ("triangle", "schauma"):
        path = os.path.join(util.GOLDEN, "_body", name + ".obj")
        v, f = cfg.load_obj(path)
        v2, f2 = gl_model.load_obj(path)
        assert np.array_equal(v, v2) and np.array_equal(f, f2)
    triangle = os.path.join(util.GOLDEN, "_body", "triangle.obj")
    assert np.array_equal(cfg.load_obj(triangle, 0.001)[0], cfg.load_obj(triangle)[0] * np.float32(0.001))


@pytest.mark.parametrize("name,region", [("region_model.bin", True), ("depth_model.bin", False)])
def test_model_file_writer_reproduces_reference_files(tmp_path, name, region):
    """write_model_bin(parse(reference file)) == reference file, byte for byte"""
    path = os.path.join(util.GOLDEN, "model_test", name)
    raw = open(path, "rb").read()
    m = gl_model.read_model_bin(path, region)
    bd, _ = cfg.BodyData.unpack(raw, 30)
    params = {k: m[k] for k in ("sphere_radius", "n_divides", "n_points", "max_radius_depth_offset",
                                "stride_depth_offset", "use_random_seed", "image_size")}
    out = tmp_path / name
    cfg.write_model_bin(str(out), region, params, bd, m["points"], m["orientations"], m["extents"])
    assert open(out, "rb").read() == raw
    assert cfg.model_bin_matches(str(out), region, params, bd)
    assert not cfg.model_bin_matches(str(out), region, dict(params, n_points=params["n_points"] + 1), bd)
    assert not cfg.model_bin_matches(str(out), not region, params, bd)
    other = cfg.BodyData(bd.geometry_path, bd.geometry_unit_in_meter, bd.geometry_counterclockwise,
                         bd.geometry_enable_culling, bd.maximum_body_diameter * 2, bd.geometry2body_pose)
    assert not cfg.model_bin_matches(str(out), region, params, other)


def test_loader_camera_naming_and_decode(tmp_path):
    api = util.open_oracle()
    cam = generator.LoaderColorCamera.from_metafile(api, os.path.join(util.GOLDEN, "_sequence", "color_camera.yaml"))
    assert os.path.basename(cam.image_path()) == "color_camera_image_200.png"
    assert cam.UpdateImage() and cam.load_index == 201
    assert np.array_equal(cam.image, util.load_color_frame(200))
    assert cam.UpdateImage() and not cam.UpdateImage()  # 201 exists, 202 does not
    ycb = generator.LoaderColorCamera(api, "/data/0048", (1, 1, 0, 0, 4, 4), "", 1, 6, "-color")
    assert ycb.image_path() == "/data/0048/000001-color.png"  # ycb_evaluator.cpp:381-382
    ycb.set_load_index(1234567)
    assert ycb.image_path() == "/data/0048/1234567-color.png"
    depth = generator.LoaderDepthCamera.from_metafile(api, os.path.join(util.GOLDEN, "_sequence", "depth_camera.yaml"))
    assert depth.depth_scale == pytest.approx(0.001) and depth.UpdateImage()
    assert depth.image.dtype == np.uint16 and np.array_equal(depth.image, util.load_depth_frame(200))


def run_generated_tracker(api, root):
    tracker = generator.GenerateConfiguredTracker(api, str(root / "tracker_test" / "tracker_config.yaml"))
    assert tracker.n_corr_iterations == 7 and tracker.n_update_iterations == 2  # tracker_test/tracker.yaml
    assert tracker.ignored == ["normal_viewer"]
    assert not tracker.DetectPoses({"triangle_optimizer"})  # not set up (tracker.cpp:209-214)
    assert tracker.SetUp()
    assert tracker.DetectPoses({"triangle_optimizer"})
    assert tracker.StartModalities(0)
    assert tracker.ExecuteTrackingStep(0)
    return tracker


def test_generated_tracker_reproduces_the_reference_pose(tmp_path):
    """TrackerTest.OptimizePoseMatrixGeneratorSetUp with the reference's criterion"""
    root = reference_tree(tmp_path)
    write_fixture_models(root)
    tracker = run_generated_tracker(util.open_oracle(), root)
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")
    pose = tracker.body_ptrs()[0].body2world_pose()
    assert np.max(np.abs((pose - golden)[:3] / golden[:3])) < 1e-5  # CompareToLoadedMatrix(..., 1.0e-5f)
    body = tracker.objects["Body"]["triangle"]
    assert (body.body_id, body.region_id) == (150, 150)
    assert tracker.objects["DepthModality"]["triangle_depth_modality"].params.measure_occlusions == 1
    assert tracker.objects["RegionModality"]["triangle_region_modality"].params.measure_occlusions == 1


def test_generator_errors(tmp_path):
    root = reference_tree(tmp_path)
    write_fixture_models(root)
    config = root / "tracker_test" / "tracker_config.yaml"
    text = config.read_text()
    api = util.open_oracle()

    def generate(new_text):
        config.write_text(new_text)
        return generator.GenerateConfiguredTracker(api, str(config))

    with pytest.raises(ValueError, match='Required parameter "root_link"'):
        generate(text.replace('    root_link: "triangle_link"\n', ""))
    with pytest.raises(ValueError, match="no_such_body"):
        generate(text.replace('bodies: ["triangle"]', 'bodies: ["no_such_body"]'))
    with pytest.raises(ValueError, match="No tracker was configured"):
        generate(text.replace("Tracker:", "Trackers:"))
    with pytest.raises(ValueError, match="TextureModality"):
        generate(text + '\nTextureModality:\n  - name: "t"\n')
    with pytest.raises(FileNotFoundError):
        generate(text.replace("../_body/triangle.yaml", "../_body/missing.yaml"))
    with pytest.raises(ValueError, match="associated bodies"):
        generate(text.replace('    body: "triangle"\n\nDepthModel',
                              '    body: "triangle"\n    fixed_bodies: ["triangle"]\n\nDepthModel'))
    # a model file generated with other parameters is not accepted: the oracle context cannot generate one
    (root / "_body" / "triangle_region_model.yaml").write_text(
        '%YAML:1.2\nmodel_path: "../../temp/triangle_region_model.bin"\nn_points: 100\n')
    with pytest.raises(util.pkg.M3TError):
        generate(text)


# ---------------------------------------------------------------------------------------------------------
# a configuration with everything the generator wires: two bodies in a kinematic tree, a focused depth renderer
# for modelled occlusions, shared colour histograms, a constraint and a soft constraint, link / optimizer /
# modality metafiles — against the same object graph built by hand
# ---------------------------------------------------------------------------------------------------------
TREE_CONFIG = """%YAML:1.2
LoaderColorCamera:
  - name: "color_camera"
    metafile_path: "../_sequence/color_camera.yaml"
LoaderDepthCamera:
  - name: "depth_camera"
    metafile_path: "../_sequence/depth_camera.yaml"
Body:
  - name: "triangle"
    metafile_path: "../_body/triangle.yaml"
  - name: "schauma"
    metafile_path: "../_body/schauma.yaml"
ColorHistograms:
  - name: "shared_histograms"
    metafile_path: "histograms.yaml"
RendererGeometry:
  - name: "renderer_geometry"
    bodies: ["triangle", "schauma"]
FocusedBasicDepthRenderer:
  - name: "depth_renderer"
    metafile_path: "renderer.yaml"
    renderer_geometry: "renderer_geometry"
    camera: "color_camera"
    referenced_bodies: ["triangle"]
RegionModel:
  - name: "triangle_region_model"
    metafile_path: "../_body/triangle_region_model.yaml"
    body: "triangle"
  - name: "schauma_region_model"
    metafile_path: "schauma_region_model.yaml"
    body: "schauma"
DepthModel:
  - name: "triangle_depth_model"
    metafile_path: "../_body/triangle_depth_model.yaml"
    body: "triangle"
RegionModality:
  - name: "triangle_region_modality"
    metafile_path: "region_modality.yaml"
    body: "triangle"
    color_camera: "color_camera"
    region_model: "triangle_region_model"
    model_occlusions: {focused_depth_renderer: "depth_renderer"}
    use_shared_color_histograms: {color_histograms: "shared_histograms"}
  - name: "schauma_region_modality"
    body: "schauma"
    color_camera: "color_camera"
    region_model: "schauma_region_model"
    use_shared_color_histograms: {color_histograms: "shared_histograms"}
DepthModality:
  - name: "triangle_depth_modality"
    body: "triangle"
    depth_camera: "depth_camera"
    depth_model: "triangle_depth_model"
Link:
  - name: "schauma_link"
    metafile_path: "schauma_link.yaml"
    body: "schauma"
    modalities: ["schauma_region_modality"]
  - name: "triangle_link"
    body: "triangle"
    modalities: ["triangle_region_modality", "triangle_depth_modality"]
    child_links: ["schauma_link"]
Constraint:
  - name: "hinge"
    metafile_path: "constraint.yaml"
    link1: "triangle_link"
    link2: "schauma_link"
SoftConstraint:
  - name: "spring"
    metafile_path: "soft_constraint.yaml"
    link1: "triangle_link"
    link2: "schauma_link"
Optimizer:
  - name: "optimizer"
    metafile_path: "optimizer.yaml"
    root_link: "triangle_link"
    constraints: ["hinge"]
    soft_constraints: ["spring"]
StaticDetector:
  - name: "detector"
    metafile_path: "../_body/triangle_static_detector.yaml"
    optimizer: "optimizer"
Tracker:
  - name: "tracker"
    metafile_path: "tracker.yaml"
    optimizers: ["optimizer"]
    detectors: ["detector"]
"""
JOINT2PARENT = np.array([[1, 0, 0, 0.17], [0, 1, 0, 0.02], [0, 0, 1, 0.09], [0, 0, 0, 1]], np.float32)


def _matrix_yaml(key, m):
    return "%s: !!opencv-matrix\n  rows: 4\n  cols: 4\n  dt: d\n  data: [ %s ]\n" % (
        key, ", ".join("%.9g" % v for v in np.asarray(m, np.float64).reshape(-1)))


def write_tree_config(root):
    t = root / "tracker_test"
    (t / "tree_config.yaml").write_text(TREE_CONFIG)
    (t / "histograms.yaml").write_text("%YAML:1.2\nn_bins: 32\nlearning_rate_f: 0.3\nlearning_rate_b: 0.1\n")
    (t / "renderer.yaml").write_text("%YAML:1.2\nimage_size: 160\nz_min: 0.05\nz_max: 5.0\n")
    (t / "region_modality.yaml").write_text("%YAML:1.2\nn_lines_max: 120\nscales: [5, 2, 1]\n"
                                            "standard_deviations: [20.0, 7.0, 3.0]\nn_unoccluded_iterations: 0\n")
    (t / "schauma_link.yaml").write_text("%YAML:1.2\n" + _matrix_yaml("joint2parent_pose", JOINT2PARENT) +
                                         "free_directions: [0, 0, 1, 1, 1, 0]\n")
    (t / "constraint.yaml").write_text("%YAML:1.2\nconstraint_directions: [1, 1, 0, 0, 0, 0]\n")
    (t / "soft_constraint.yaml").write_text("%YAML:1.2\nconstraint_directions: [0, 0, 0, 1, 1, 1]\n"
                                            "max_distance_translation: 0.002\nstandard_deviation_translation: 0.005\n")
    (t / "optimizer.yaml").write_text("%YAML:1.2\ntikhonov_parameter_rotation: 5000\n"
                                      "tikhonov_parameter_translation: 200000\n")
    (t / "schauma_region_model.yaml").write_text(
        '%YAML:1.2\nmodel_path: "../../temp/schauma_region_model.bin"\nsphere_radius: 0.4\nn_divides: 2\n'
        'n_points: 10\nmax_radius_depth_offset: 0.05\nstride_depth_offset: 0.002\nimage_size: 500\n')
    # the reference's own region model of the bottle, re-headed for the mesh file of this tree
    src = os.path.join(util.GOLDEN, "model_test", "region_model.bin")
    m = gl_model.read_model_bin(src, True)
    body_yaml = cfg.read_yaml(str(root / "_body" / "schauma.yaml"))
    v, _ = cfg.load_obj(str(root / "_body" / "schauma.obj"), body_yaml["geometry_unit_in_meter"])
    g2b = cfg.pose(body_yaml["geometry2body_pose"])
    bd = cfg.BodyData(str(root / "_body" / "schauma.obj"), body_yaml["geometry_unit_in_meter"],
                      body_yaml["geometry_counterclockwise"], body_yaml["geometry_enable_culling"],
                      cfg.maximum_body_diameter(v @ g2b[:3, :3].T + g2b[:3, 3]), g2b)
    params = {k: m[k] for k in ("sphere_radius", "n_divides", "n_points", "max_radius_depth_offset",
                                "stride_depth_offset", "use_random_seed", "image_size")}
    cfg.write_model_bin(str(root.parent / "temp" / "schauma_region_model.bin"), True, params, bd, m["points"],
                        m["orientations"], m["extents"])
    return m


def build_tree_by_hand(api, root, schauma_model):
    """the object graph of TREE_CONFIG through the host classes, in the generator's creation order"""
    h = util.host
    ty = cfg.read_yaml(str(root / "_body" / "triangle.yaml"))
    sy = cfg.read_yaml(str(root / "_body" / "schauma.yaml"))
    triangle, schauma = h.Body(api), h.Body(api)
    tv, tf = cfg.load_obj(str(root / "_body" / "triangle.obj"), ty["geometry_unit_in_meter"])
    sv, sf = cfg.load_obj(str(root / "_body" / "schauma.obj"), sy["geometry_unit_in_meter"])
    triangle.set_geometry(tv, tf, cfg.pose(ty["geometry2body_pose"]), True, True, ty["body_id"], ty["region_id"])
    schauma.set_geometry(sv, sf, cfg.pose(sy["geometry2body_pose"]), bool(sy["geometry_counterclockwise"]),
                         bool(sy["geometry_enable_culling"]), sy["body_id"], sy["region_id"])
    histograms = h.ColorHistograms(api, 32, 0.3, 0.1)
    geometry = h.RendererGeometry(api)
    geometry.AddBody(triangle)
    geometry.AddBody(schauma)
    color = h.ColorCamera(api, **util.COLOR_INTR)
    # (camera2world is a float transform in the reference: rounded to f32 before it is inverted)
    depth = h.DepthCamera(api, depth_scale=0.001,
                          world2camera_pose=generator._inverse_pose(cfg.pose(util.DEPTH_CAMERA2WORLD)), **util.DEPTH_INTR)
    renderer = h.FocusedBasicDepthRenderer(api, geometry, color, image_size=160, z_min=0.05, z_max=5.0)
    renderer.AddReferencedBody(triangle)
    v = gs.views()
    t_region = h.RegionModel(api, data_points=v["region_points"], orientations=v["region_orientations"],
                             contour_lengths=v["region_contour_lengths"])
    s_region = h.RegionModel(api, data_points=schauma_model["points"], orientations=schauma_model["orientations"],
                             contour_lengths=schauma_model["extents"],
                             stride_depth_offset=schauma_model["stride_depth_offset"],
                             max_radius_depth_offset=schauma_model["max_radius_depth_offset"])
    t_depth = h.DepthModel(api, data_points=v["depth_points"], orientations=v["depth_orientations"],
                           surface_areas=v["depth_surface_areas"])
    rm_t = h.RegionModality(api, triangle, color, t_region, n_lines_max=120, scales=[5, 2, 1],
                            standard_deviations=[20.0, 7.0, 3.0], n_unoccluded_iterations=0)
    rm_t.ModelOcclusions(renderer)
    rm_t.UseSharedColorHistograms(histograms)
    rm_s = h.RegionModality(api, schauma, color, s_region)
    rm_s.UseSharedColorHistograms(histograms)
    dm_t = h.DepthModality(api, triangle, depth, t_depth)
    link_t = h.Link(api, triangle)
    link_t.AddModality(rm_t)
    link_t.AddModality(dm_t)
    link_s = h.Link(api, schauma, link_t, np.eye(4), JOINT2PARENT, (0, 0, 1, 1, 1, 0))
    link_s.AddModality(rm_s)
    optimizer = h.Optimizer(api, link_t, tikhonov_parameter_rotation=5000.0, tikhonov_parameter_translation=200000.0)
    h.Constraint(api, optimizer, link_t, link_s, constraint_directions=(1, 1, 0, 0, 0, 0))
    h.SoftConstraint(api, optimizer, link_t, link_s, constraint_directions=(0, 0, 0, 1, 1, 1),
                     max_distance_translation=0.002, standard_deviation_translation=0.005)
    tracker = h.Tracker(api, 4, 2)
    color.UpdateImage(util.load_color_frame(200))
    depth.UpdateImage(util.load_depth_frame(200))
    detector = cfg.pose(cfg.read_yaml(str(root / "_body" / "triangle_static_detector.yaml"))["link2world_pose"])
    link_t.set_link2world_pose(detector)
    assert tracker.CalculateConsistentPoses()
    return tracker, (triangle, schauma), (rm_t, rm_s)


def test_generator_wires_trees_renderers_shared_histograms_and_constraints(tmp_path):
    root = reference_tree(tmp_path)
    write_fixture_models(root)
    schauma_model = write_tree_config(root)
    (root / "tracker_test" / "tracker.yaml").write_text("%YAML:1.2\nn_corr_iterations: 4\nn_update_iterations: 2\n")

    generated = generator.GenerateConfiguredTracker(util.open_oracle(), str(root / "tracker_test" / "tree_config.yaml"))
    assert generated.SetUp() and generated.DetectPoses({"optimizer"})
    assert generated.StartModalities(0) and generated.ExecuteTrackingStep(0)

    tracker, bodies, region = build_tree_by_hand(util.open_oracle(), root, schauma_model)
    assert tracker.StartModalities(0) and tracker.ExecuteTrackingStep(0)

    for name, body in zip(("triangle", "schauma"), bodies):
        assert np.array_equal(generated.objects["Body"][name].body2world_pose(), body.body2world_pose()), name
    for name, modality in zip(("triangle_region_modality", "schauma_region_modality"), region):
        a = generated.objects["RegionModality"][name].histograms()
        b = modality.histograms()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # the child hangs on its parent through the joint: free are the rotation about z and the translations along
    # x and y (schauma_link.yaml), so the relative pose keeps the joint's height and its z axis
    t, s = (b.body2world_pose() for b in bodies)
    relative = np.linalg.inv(t.astype(np.float64)) @ s
    assert abs(relative[2, 3] - JOINT2PARENT[2, 3]) < 1e-5
    assert np.allclose(relative[:3, 2], [0, 0, 1], atol=1e-5) and np.allclose(relative[2, :3], [0, 0, 1], atol=1e-5)
    assert generated.objects["Link"]["schauma_link"].parent is generated.objects["Link"]["triangle_link"]
    assert generated.objects["FocusedBasicDepthRenderer"]["depth_renderer"].image_size == 160


def test_static_detector_golden(tmp_path):
    """StaticDetectorTest.DetectPose (test/detector_test.cpp:77-85): the detector's metafile pose lands on the body
    exactly (CompareToLoadedMatrix(..., 0.0f)); a name that is not the detector's optimizer leaves it alone"""
    root = reference_tree(tmp_path)
    write_fixture_models(root)
    tracker = generator.GenerateConfiguredTracker(util.open_oracle(), str(root / "tracker_test" / "tracker_config.yaml"))
    assert tracker.SetUp()
    body = tracker.body_ptrs()[0]
    before = body.body2world_pose()
    assert tracker.DetectPoses({"some_other_optimizer"})
    assert np.array_equal(body.body2world_pose(), before)
    detected = set()
    assert tracker.DetectPoses({"triangle_optimizer"}, detected) and detected == {"triangle_optimizer"}
    golden = util.read_golden_matrix("detector_test/detector_triangle_pose.txt").astype(np.float32)
    assert np.array_equal(body.body2world_pose(), golden)
    link = tracker.objects["Link"]["triangle_link"]
    assert np.array_equal(link.link2world_pose(), golden)


def test_demo_configuration_is_read_and_refused_by_name():
    """data/pen_paper_demo/config.yaml: unquoted scalars, comments after values, flow sequences; it asks for sensor
    cameras and a texture modality, which the generator names when it refuses"""
    path = os.path.join(util.GOLDEN, "pen_paper_demo", "config.yaml")
    d = cfg.read_yaml(path)
    assert [b["name"] for b in d["Body"]] == ["stabilo", "stabilo_body", "stabilo_tip", "paper"]
    assert d["Link"][0]["modalities"] == ["stabilo_tip_region_modality", "stabilo_body_region_modality",
                                          "stabilo_texture_modality", "stabilo_depth_modality"]
    assert d["RegionModel"][0]["fixed_bodies"] == ["stabilo_body"]
    assert d["RegionModality"][2]["measure_occlusions"] == {"depth_camera": "depth_camera"}
    m = cfg.read_yaml(os.path.join(util.GOLDEN, "pen_paper_demo", "stabilo_region_modality.yaml"))
    assert m == {"use_adaptive_coverage": 1, "reference_contour_length": 0.23, "measured_occlusion_threshold": 0.01}
    with pytest.raises(ValueError, match="TextureModality|AzureKinect"):
        generator.GenerateConfiguredTracker(util.open_oracle(), path)
