"""The claim behind ldlt_solve_rows' pivot order (csrc/m3t_links.hip): for distinct |diagonal| values Eigen's
step-by-step selection with swaps ends with "position p holds the row of rank p"; with ties it does not (the kernel
detects ties and takes the step-by-step selection then)."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rank_order_equals_selection_order_for_distinct_diagonals(tmp_path):
    exe = str(tmp_path / "pivot_order_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "cpp", "pivot_order_check.cpp")],
                   check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    m = re.match(r"cases (\d+) mismatches (\d+) ties_that_differ (\d+)", out.stdout)
    assert m and out.returncode == 0, out.stdout + out.stderr
    cases, mismatches, ties = map(int, m.groups())
    assert cases > 100000 and mismatches == 0 and ties > 0
