"""3dobjecttracking_amd/csrc/m3t_raster.h on the host: the row scan the focused renderers' kernels use (three additions
per pixel, a row left behind its covered span; serial and in the 32-pixel pieces of the workgroup path) against the
per-pixel definition (three edge functions, top-left rule) on random triangles -- slivers, vertices on pixel centres
and pixel edges, horizontal edges, triangles that leave the image: the same packed depth word for every pixel."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_scan_equals_the_per_pixel_definition(tmp_path):
    exe = str(tmp_path / "raster_check")
    subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "raster_check.cpp")], check=True)
    out = subprocess.run([exe, "40000"], capture_output=True, text=True, timeout=600)
    m = re.match(r"triangles (\d+) covered (\d+) mismatches (\d+)", out.stdout)
    assert m and out.returncode == 0, out.stdout + out.stderr
    triangles, covered, mismatches = map(int, m.groups())
    assert triangles > 30000 and covered > 10 ** 7 and mismatches == 0
