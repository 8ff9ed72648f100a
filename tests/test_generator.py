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
