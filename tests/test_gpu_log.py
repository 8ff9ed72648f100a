"""The device's logarithm (csrc/m3t_log.h + ocml's double logarithm where the table path does not vouch for its
rounding; region_modality.cpp:520-523 is the call site) against the oracle's expression float(std::log(double(x)))
evaluated by glibc on the host, for EVERY float in [FLT_MIN, 1]: position-weighted checksum of the result bits and the
number of evaluations that took the general logarithm.  Stated tolerance: none."""
import ctypes as C
import os
import re
import subprocess

import pytest

import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_device_logarithm_equals_the_host_expression_on_every_float(tmp_path):
    exe = str(tmp_path / "log_check")
    subprocess.run(["g++", "-O2", "-march=x86-64-v3", "-ffp-contract=off", "-std=c++17", "-fopenmp", "-o", exe,
                    os.path.join(ROOT, "tests", "cpp", "log_check.cpp")], check=True)
    out = subprocess.run([exe, "1"], capture_output=True, text=True, timeout=900)
    api = util.open_hip()
    got = (C.c_ulonglong * 3)()
    for first, last in ((0x00800000, 0x3F800000), (0x3F000000, 0x3F800000), (0x00800000, 0x00FFFFFF)):
        out = subprocess.run([exe, "1", str(first), str(last)], capture_output=True, text=True, timeout=900)
        m = re.match(r"checked (\d+) mismatches (\d+) fallbacks (\d+) wrongly_taken (\d+) checksum (\d+) "
                     r"fallback_checksum (\d+)", out.stdout)
        assert m and out.returncode == 0, out.stdout + out.stderr
        checked, mismatches, fallbacks, _, checksum, fallback_checksum = map(int, m.groups())
        assert checked == last - first + 1 and mismatches == 0
        api.call("debug_log_checksum", first, last, got)
        assert got[1] == fallbacks  # the same inputs leave the table path on both sides ...
        assert got[2] == fallback_checksum  # ... ocml's double logarithm rounds to glibc's f32 on each of them ...
        assert got[0] == checksum  # ... and every result has the host's bits
    assert fallbacks > 0
