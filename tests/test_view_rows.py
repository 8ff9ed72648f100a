"""closest_view_local's exactness (3dobjecttracking_amd/csrc/m3t_view_rows.h, m3t_kernels.hip) on the host: whenever
the previous view's row vouches for a viewing direction, the arg-max over the row is the arg-max of
RegionModel::GetClosestView's scan over all views (region_model.cpp:105-130) -- for the 2562-view geodesic set of the
benchmark models, for the reference's 162-view golden model, and for a random (non-uniform) view set."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import util

sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
import gl_model  # noqa: E402  (reader of the reference's .bin models)

ROOT = util.ROOT


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("view_rows") / "view_rows_check")
    subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "view_rows_check.cpp")], check=True)
    return exe


def run(checker, tmp_path, orientations, n_directions):
    path = tmp_path / "views.f32"
    np.ascontiguousarray(orientations, np.float32).tofile(path)
    out = subprocess.run([checker, str(path), str(len(orientations)), str(n_directions)], capture_output=True, text=True,
                         timeout=900)
    m = re.match(r"directions (\d+) vouched (\d+) mismatches (\d+)", out.stdout)
    assert m and out.returncode == 0, out.stdout + out.stderr
    return [int(x) for x in m.groups()]


def test_geodesic_2562_views(checker, tmp_path):
    ori = util.syn.geodesic_points(4)
    assert ori.shape == (2562, 3)
    n, vouched, mismatches = run(checker, tmp_path, ori, 60000)
    assert mismatches == 0
    assert 0.5 * n < vouched < n  # both outcomes occur: the row vouches for small moves, not for large ones


def test_reference_golden_model_views(checker, tmp_path):
    views = gl_model.read_model_bin(os.path.join(util.GOLDEN, "model_test", "region_model.bin"), True)
    ori = np.asarray(views["orientations"], np.float32)
    assert ori.shape == (162, 3)
    n, vouched, mismatches = run(checker, tmp_path, ori, 60000)
    assert mismatches == 0 and vouched > 0.3 * n


def test_random_view_set(checker, tmp_path):
    rng = np.random.default_rng(5)
    ori = rng.normal(size=(700, 3))
    ori /= np.linalg.norm(ori, axis=1, keepdims=True)
    ori[100] = ori[99]  # a duplicated view: equal dot products, the lower index must win
    n, vouched, mismatches = run(checker, tmp_path, ori, 60000)
    assert mismatches == 0 and vouched > 0
