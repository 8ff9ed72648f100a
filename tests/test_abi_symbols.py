"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/m3t_hip.h declares (same for the oracle header), and fails loudly without a GPU."""
import ctypes
import os
import re

import pytest

import util

ROOT = util.ROOT


def _declared(header, prefix):
    txt = open(os.path.join(ROOT, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_hip_library_exports_every_declared_symbol():
    names = _declared("include/m3t_hip.h", "m3t_hip_")
    assert len(names) >= 45
    assert os.path.exists(util.pkg.LIB_PATH), "build first: python __graft_entry__.py"
    lib = ctypes.CDLL(util.pkg.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # the reference's entry-point names for the path (M3T ExecuteTrackingStep / ICG ExecuteTrackingCycle)
    assert "m3t_hip_execute_tracking_step" in names and "m3t_hip_execute_tracking_cycle" in names


def test_oracle_library_exports_every_declared_symbol():
    names = _declared("oracle/m3t_oracle.h", "m3t_oracle_")
    lib = ctypes.CDLL(util.build_oracle())
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # both sides expose the same tracker / modality surface
    hip = {n[len("m3t_hip_"):] for n in _declared("include/m3t_hip.h", "m3t_hip_")}
    ora = {n[len("m3t_oracle_"):] for n in names if "histograms_" not in n or "modality" in n}
    assert ora <= hip, sorted(ora - hip)


def test_python_signatures_cover_the_header():
    capi = util.pkg._capi
    declared = {n[len("m3t_hip_"):] for n in _declared("include/m3t_hip.h", "m3t_hip_")}
    bound = set(capi._SIGNATURES) | set(capi._HIP_ONLY) | {"create", "destroy", "last_error"}
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))


def test_struct_layouts_match_the_c_header(tmp_path):
    """sizeof() of every POD crossing the boundary, C compiler vs ctypes."""
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "m3t_types.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(m3t_intrinsics),sizeof(m3t_region_model_desc),sizeof(m3t_depth_model_desc),"
                   "sizeof(m3t_region_modality_params),sizeof(m3t_depth_modality_params),sizeof(m3t_data_line),"
                   "sizeof(m3t_data_point));return 0;}\n")
    exe = tmp_path / "sz"
    import subprocess
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    sizes = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    c = util.pkg._capi
    expect = [ctypes.sizeof(t) for t in (c.Intrinsics, c.RegionModelDesc, c.DepthModelDesc, c.RegionModalityParams,
                                         c.DepthModalityParams, c.DataLine, c.DataPoint)]
    assert sizes == expect
    # python defaults == header defaults (region_modality.h:411-443, depth_modality.h:302-321)
    p = c.RegionModalityParams()
    assert (p.n_lines_max, p.function_length, p.distribution_length, list(p.scales)[:4]) == (200, 8, 12, [6, 4, 2, 1])
    d = c.DepthModalityParams()
    assert d.n_points_max == 200 and abs(d.stride_length - 0.005) < 1e-9


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a box without a GPU")
def test_product_fails_loudly_without_gpu():
    with pytest.raises(util.pkg.M3TError) as e:
        util.pkg.open_context(0)
    assert e.value.code == util.pkg._capi.M3T_ERR_DEVICE
    assert "HIP device" in str(e.value)
