"""include/m3t_hip_modality.h — the m3t::Modality adapter of INTEGRATION.md §2 — compiled against interface stubs
of the reference's headers (tests/cpp/m3t_stub/) and driven by a host that only knows m3t::Modality
(tests/cpp/adapter_demo.cpp).  On CPU the C-ABI names are mapped onto the oracle library, so the adapter's own
logic (shared batch, one launch per round, pose push, image upload, g/H copy-back) runs everywhere; on the GPU the
same program links libm3t_hip.so.  Either way its gradients / Hessians must equal those of the same scene driven
through the Python host, bit for bit."""
import os
import re
import subprocess

import numpy as np
import pytest

import golden_scene as gs
import util
from util import host

ROOT = util.ROOT
cfg = util.pkg.config
generator = util.pkg.generator
INCLUDES = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp", "m3t_stub")]
SRC = os.path.join(ROOT, "tests", "cpp", "adapter_demo.cpp")


def write_scene(directory):
    v = gs.views()
    params = dict(generator._MODEL_DEFAULTS)
    data = cfg.BodyData("triangle.obj", 1.0, True, True, 0.1, np.eye(4))
    cfg.write_model_bin(str(directory / "region.bin"), True, params, data, v["region_points"], v["region_orientations"],
                        v["region_contour_lengths"])
    cfg.write_model_bin(str(directory / "depth.bin"), False, params, data, v["depth_points"], v["depth_orientations"],
                        v["depth_surface_areas"])
    util.load_color_frame(200).tofile(directory / "color.raw")
    util.load_depth_frame(200).tofile(directory / "depth.raw")
    world2camera = generator._inverse_pose(cfg.pose(util.DEPTH_CAMERA2WORLD))
    body2world = gs.mtv.body2world()
    c, d = util.COLOR_INTR, util.DEPTH_INTR
    numbers = [c["fu"], c["fv"], c["ppu"], c["ppv"], c["width"], c["height"],
               d["fu"], d["fv"], d["ppu"], d["ppv"], d["width"], d["height"], 0.001]
    numbers += list(np.asarray(world2camera, np.float32).T.reshape(-1)) + list(np.asarray(body2world, np.float32).T.reshape(-1))
    (directory / "scene.txt").write_text(" ".join("%.9g" % x for x in numbers) + "\n")
    return world2camera, body2world


def expected(api, directory, world2camera, body2world):
    """the same object graph through the Python host (no Link / Optimizer: the host keeps those in adapter mode)"""
    body = host.Body(api, body2world)
    color = host.ColorCamera(api, **util.COLOR_INTR)
    depth = host.DepthCamera(api, depth_scale=0.001, world2camera_pose=world2camera, **util.DEPTH_INTR)
    region = host.RegionModality(api, body, color, host.RegionModel(api, path=str(directory / "region.bin")),
                                 depth_camera=depth, measure_occlusions=1)
    depth_modality = host.DepthModality(api, body, depth, host.DepthModel(api, path=str(directory / "depth.bin")),
                                        measure_occlusions=1)
    tracker = host.Tracker(api)
    if api.is_hip:
        api.call("set_fused_step", 0)
    color.UpdateImage(util.load_color_frame(200))
    depth.UpdateImage(util.load_depth_frame(200))
    assert tracker.StartModalities(0) and tracker.CalculateCorrespondences(0, 0)
    assert tracker.CalculateGradientAndHessian(0, 0, 0)
    first = [np.concatenate([m.gradient(), m.hessian().T.reshape(-1)]) for m in (region, depth_modality)]
    moved = np.array(body2world, np.float32)
    moved[0, 3] += np.float32(0.001)
    body.set_body2world_pose(moved)
    assert tracker.CalculateGradientAndHessian(0, 0, 1)
    return first, depth_modality.gradient()


def expected_fast_pose(api, directory, world2camera, body2world):
    """TrackerTest.OptimizePoseMatrix's object graph: StartModalities + one ExecuteTrackingStep of 7 x 2 iterations"""
    body = host.Body(api, body2world)
    color = host.ColorCamera(api, **util.COLOR_INTR)
    depth = host.DepthCamera(api, depth_scale=0.001, world2camera_pose=world2camera, **util.DEPTH_INTR)
    region = host.RegionModality(api, body, color, host.RegionModel(api, path=str(directory / "region.bin")),
                                 depth_camera=depth, measure_occlusions=1)
    depth_modality = host.DepthModality(api, body, depth, host.DepthModel(api, path=str(directory / "depth.bin")),
                                        measure_occlusions=1)
    host.Optimizer(api, body=body, modalities=[region, depth_modality])
    tracker = host.Tracker(api, 7, 2)
    color.UpdateImage(util.load_color_frame(200))
    depth.UpdateImage(util.load_depth_frame(200))
    assert tracker.StartModalities(0) and tracker.ExecuteTrackingStep(0)
    return body.body2world_pose()


def run_demo(exe, directory, *args):
    out = subprocess.run([exe, str(directory)] + list(args), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "Set up modality triangle_region_modality first" in out.stderr
    rows = {line.split()[0]: np.array([float.fromhex(x) for x in line.split()[1:]], np.float32)
            for line in out.stdout.strip().splitlines()}
    fast = rows["fast"].reshape(4, 4).T if "fast" in rows else None
    if fast is not None:
        # device-optimisation mode: the unmodified host loop (tracker.cpp:344-364) over the adapters ends on the same pose
        # as the library's own ExecuteTrackingStep, bit for bit
        assert np.array_equal(rows["fused"].reshape(4, 4).T, fast)
    # a repeated round (same iteration indices after a reset) ran again and reproduced the first result
    assert np.array_equal(rows["again"], rows["triangle_region_modality"])
    return [rows["triangle_region_modality"], rows["triangle_depth_modality"]], rows["moved"], fast


def test_adapter_header_compiles_against_the_interface_stubs():
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only"] + INCLUDES + [SRC])


def test_adapter_over_the_oracle_library(tmp_path):
    """the adapter's logic on CPU: every m3t_hip_* name of the header is mapped onto its m3t_oracle_* twin"""
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "m3t_hip.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(m3t_hip_\w+)\s*\(", header)))
    rename = tmp_path / "rename.h"
    rename.write_text("#define m3t_hip_context m3t_oracle_context\n" +
                      "".join("#define %s %s\n" % (n, n.replace("m3t_hip_", "m3t_oracle_")) for n in names))
    shim = tmp_path / "shim.cpp"  # the one entry point of the adapter the oracle has no use for
    shim.write_text('struct m3t_oracle_context;\nextern "C" int m3t_oracle_set_fused_step(m3t_oracle_context*, int) '
                    "{ return 0; }\n")
    util.build_oracle()
    exe = str(tmp_path / "adapter_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-include", str(rename)] + INCLUDES +
                          [SRC, str(shim), "-o", exe, "-L", util.ORACLE_DIR, "-lm3t_oracle",
                           "-Wl,-rpath," + util.ORACLE_DIR])
    world2camera, body2world = write_scene(tmp_path)
    got, got_moved, fast_pose = run_demo(exe, tmp_path)
    want, want_moved = expected(util.open_oracle(), tmp_path, world2camera, body2world)
    for a, b in zip(got, want):
        assert a.shape == (42,) and np.array_equal(a, b)
    assert np.abs(got[0]).max() > 0 and np.abs(got[1]).max() > 0
    assert np.array_equal(got_moved, want_moved) and not np.array_equal(got_moved, want[1][:6])
    # fast mode: the host Body ends on the pose the Python-driven tracker reaches, which is the reference's golden
    assert np.array_equal(fast_pose, expected_fast_pose(util.open_oracle(), tmp_path, world2camera, body2world))
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")
    assert np.max(np.abs((fast_pose - golden)[:3] / golden[:3])) < 1e-5


@pytest.mark.gpu
def test_adapter_over_the_hip_library(tmp_path):
    libdir = os.path.dirname(util.pkg.LIB_PATH)
    exe = str(tmp_path / "adapter_demo")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall"] + INCLUDES + [SRC, "-o", exe, "-L", libdir, "-lm3t_hip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    world2camera, body2world = write_scene(tmp_path)
    got, got_moved, fast_pose = run_demo(exe, tmp_path)
    want, want_moved = expected(util.open_hip(), tmp_path, world2camera, body2world)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert np.array_equal(got_moved, want_moved)
    # fast mode and device-optimisation mode (run_demo checks that the two agree): the pose of the Python-driven
    # tracker on the same library, which meets the reference's TrackerTest golden
    assert np.array_equal(fast_pose, expected_fast_pose(util.open_hip(), tmp_path, world2camera, body2world))
    golden = util.read_golden_matrix("tracker_test/triangle_pose.txt")
    assert np.max(np.abs((fast_pose - golden)[:3] / golden[:3])) < 1e-5
