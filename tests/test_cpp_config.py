"""include/m3t_hip_config.hpp, the C++ configuration front-end: YAML metafiles, meshes, PNG frames and model
files against the Python readers / the reference's own files on CPU; on the GPU the reference's
tracker_config.yaml end to end against the golden pose of TrackerTest.OptimizePoseMatrixGeneratorSetUp."""
import os
import subprocess

import numpy as np
import pytest

import util
from test_generator import reference_tree

ROOT = util.ROOT
SRC = os.path.join(ROOT, "tests", "cpp", "config_demo.cpp")
cfg = util.pkg.config


@pytest.fixture(scope="module")
def demo(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cpp_config") / "config_demo")
    libdir = os.path.dirname(util.pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           "-o", exe, "-L", libdir, "-lm3t_hip", "-lz", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])

    def run(*args, ok=True):
        out = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
        assert (out.returncode == 0) == ok, (out.returncode, out.stdout, out.stderr)
        return out.stdout.strip() if ok else out.stderr.strip()
    return run


def test_yaml_reader_agrees_with_the_python_reader(demo):
    g = util.GOLDEN
    config = os.path.join(g, "tracker_test", "tracker_config.yaml")
    d = cfg.read_yaml(config)
    assert demo("yaml", config).split()[2:] == list(d.keys())
    assert demo("yaml", config, "RegionModality", 0, "measure_occlusions", "depth_camera") == "depth_camera"
    assert demo("yaml", config, "Link", 0, "modalities").split()[2:] == d["Link"][0]["modalities"]
    assert demo("yaml", config, "Tracker", 0, "metafile_path") == "tracker.yaml"
    assert demo("yaml", config, "measure_occlusions") == "1"
    m = demo("yaml", os.path.join(g, "_sequence", "depth_camera.yaml"), "camera2world_pose").split()
    assert m[:3] == ["matrix", "4", "4"]
    assert np.allclose([float(x) for x in m[3:]], util.DEPTH_CAMERA2WORLD.reshape(-1), atol=1e-9)
    assert demo("yaml", os.path.join(g, "_body", "triangle.yaml"), "geometry_path") == "triangle.obj"
    assert demo("yaml", os.path.join(g, "_sequence", "color_camera.yaml"), "intrinsics", "f_u") == "698.128"
    assert "Could not open file" in demo("yaml", os.path.join(g, "missing.yaml"), ok=False)


def test_mesh_png_and_model_files(demo, tmp_path):
    g = util.GOLDEN
    for name, unit in (("triangle", 1.0), ("schauma", 0.001)):
        path = os.path.join(g, "_body", name + ".obj")
        v, f = cfg.load_obj(path, unit)
        nv, nf, vs, fs, diameter = demo("obj", path, unit).split()
        assert (int(nv), int(nf)) == (len(v), len(f)) and int(fs) == int(f.astype(np.int64).sum())
        assert float(vs) == pytest.approx(float(v.astype(np.float64).sum()), rel=1e-6, abs=1e-6)
        assert float(diameter) == pytest.approx(cfg.maximum_body_diameter(v), rel=1e-6)
    for name, image in (("color_camera_image_200.png", util.load_color_frame(200)),
                        ("depth_camera_image_201.png", util.load_depth_frame(201))):
        w, h, ch, bpc, total = [int(x) for x in demo("png", os.path.join(g, "_sequence", name)).split()]
        assert (h, w) == image.shape[:2] and ch == (3 if image.ndim == 3 else 1) and bpc == image.itemsize
        assert total == int(image.view(np.uint8).astype(np.uint64).sum())
    for name, region in (("region_model.bin", 1), ("depth_model.bin", 0)):
        src = os.path.join(g, "model_test", name)
        out = tmp_path / name
        assert demo("bin", src, region, out) == "1 0"  # accepted as its own kind, refused as the other
        assert open(out, "rb").read() == open(src, "rb").read()


@pytest.mark.gpu
def test_cpp_generator_errors(demo, tmp_path):
    root = reference_tree(tmp_path)
    config = root / "tracker_test" / "tracker_config.yaml"
    text = config.read_text()
    config.write_text(text + '\nTextureModality:\n  - name: "t"\n')
    assert "TextureModality" in demo("track", config, ok=False)


@pytest.mark.gpu
def test_cpp_generated_tracker_reproduces_the_reference_pose(demo, tmp_path):
    root = reference_tree(tmp_path)
    out = demo("track", root / "tracker_test" / "tracker_config.yaml", 1).splitlines()
    name, n_corr, n_update, *pose = out[0].split()
    assert (name, n_corr, n_update) == ("triangle", "7", "2")
    pose = np.array([float.fromhex(x) for x in pose], np.float32).reshape(4, 4).T
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")
    assert np.max(np.abs((pose - golden)[:3] / golden[:3])) < 1e-5  # CompareToLoadedMatrix(..., 1.0e-5f)
    models = dict(line.split()[1:] for line in out[1:])
    assert models["triangle_region_model"] == str(tmp_path / "temp" / "triangle_region_model.bin")
    for path in models.values():
        assert os.path.getsize(path) > 2562 * 200 * 36 * 4
    # the Python front-end accepts the files the C++ front-end wrote, and lands on the same pose bit for bit
    api = util.open_hip()
    api.call("set_summation_mode", 1)
    tracker = util.pkg.generator.GenerateConfiguredTracker(api, str(root / "tracker_test" / "tracker_config.yaml"))
    stamp = os.path.getmtime(models["triangle_region_model"])
    assert tracker.SetUp() and tracker.DetectPoses({"triangle_optimizer"})
    assert tracker.StartModalities(0) and tracker.ExecuteTrackingStep(0)
    assert os.path.getmtime(models["triangle_region_model"]) == stamp
    assert np.array_equal(tracker.body_ptrs()[0].body2world_pose(), pose)
