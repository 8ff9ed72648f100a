"""The C++ host mirror (include/m3t_hip.hpp) compiles with plain g++ against the C-ABI (CPU
check), and on the GPU a C++ program driving the tracker through it reproduces the poses of the
Python-driven run bit for bit."""
import os
import subprocess

import numpy as np
import pytest

import scenes
import util

ROOT = util.ROOT
SRC = os.path.join(ROOT, "tests", "cpp", "host_demo.cpp")


def _build(tmp_path):
    exe = str(tmp_path / "host_demo")
    libdir = os.path.dirname(util.pkg.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", libdir, "-lm3t_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_mirror_compiles_and_links(tmp_path):
    exe = _build(tmp_path)
    assert os.path.exists(exe)
    # plain C translation unit: the ABI headers are C, not C++
    c = tmp_path / "abi.c"
    c.write_text('#include "m3t_hip.h"\nint main(void){m3t_region_modality_params p;'
                 'm3t_region_modality_params_default(&p);return p.n_lines_max==200?0:1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(c),
                           "-o", str(tmp_path / "abi.o")])


def _run_cpp_host(exe, tmp_path):
    inputs = scenes.Inputs(3, 3, n_divides=2)
    d = tmp_path / "scene"
    d.mkdir()
    i0 = inputs.intr
    dp0 = inputs.region_models[0][0]
    (d / "scene.txt").write_text("%d %d %d %d %r %r %r %r %d %d\n" % (
        inputs.n_objects, inputs.n_frames, i0["width"], i0["height"], i0["fu"], i0["fv"], i0["ppu"], i0["ppv"],
        dp0.shape[0], dp0.shape[1]))
    for i in range(inputs.n_objects):
        dp, ori, cl = inputs.region_models[inputs.model_of[i]]
        with open(d / ("model_%d.bin" % i), "wb") as f:
            f.write(np.ascontiguousarray(dp, np.float32).tobytes())
            f.write(np.ascontiguousarray(ori, np.float32).tobytes())
            f.write(np.ascontiguousarray(cl, np.float32).tobytes())
        np.ascontiguousarray(np.asarray(inputs.start[i], np.float32).T).tofile(d / ("start_%d.bin" % i))
        for k in range(inputs.n_frames):
            np.ascontiguousarray(inputs.color[i][k]).tofile(d / ("frame_%d_%d.bin" % (i, k)))
    out = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert "first" in out.stderr  # the reference's "Set up ... first" convention on the premature call
    cpp = np.array([[float.fromhex(x) for x in line.split()] for line in out.stdout.strip().splitlines()], np.float32)
    return inputs, cpp


def _python_poses(api, inputs):
    a = scenes.Instance(api, inputs)
    a.upload_frame(0)
    assert a.tracker.StartModalities(0)
    for k in range(inputs.n_frames):
        a.upload_frame(k)
        assert a.tracker.ExecuteTrackingStep(k)
    return np.stack([np.ascontiguousarray(p.T).reshape(16) for p in a.poses()])


def test_cpp_host_over_the_oracle_library(tmp_path):
    """the same C++ program on CPU: the C-ABI names mapped onto the oracle library (tests/test_cpp_config.py)"""
    from test_cpp_config import _oracle_mapping
    rename, shim = _oracle_mapping(tmp_path)
    util.build_oracle()
    exe = str(tmp_path / "host_demo_oracle")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-parameter", "-include", str(rename), "-I",
                           os.path.join(ROOT, "include"), SRC, str(shim), "-o", exe, "-L", util.ORACLE_DIR,
                           "-lm3t_oracle", "-Wl,-rpath," + util.ORACLE_DIR])
    inputs, cpp = _run_cpp_host(exe, tmp_path)
    assert np.array_equal(cpp, _python_poses(util.open_oracle(), inputs))


@pytest.mark.gpu
def test_cpp_host_matches_python_host(tmp_path):
    exe = _build(tmp_path)
    inputs, cpp = _run_cpp_host(exe, tmp_path)
    assert np.array_equal(cpp, _python_poses(util.open_hip(), inputs))
