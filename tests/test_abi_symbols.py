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


def test_hip_library_exports_nothing_but_the_declared_symbols():
    """the dynamic symbol table of libm3t_hip.so is the C-ABI and nothing else (-fvisibility=hidden + the version script
    csrc/libm3t_hip.map): no kernel launch stubs, no std:: instantiations that another library's copy could interpose"""
    import subprocess
    out = subprocess.check_output(["nm", "-D", "--defined-only", util.pkg.LIB_PATH], text=True)
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared("include/m3t_hip.h", "m3t_hip_"), sorted(set(exported) ^ set(_declared("include/m3t_hip.h", "m3t_hip_")))


def test_oracle_library_exports_every_declared_symbol():
    names = _declared("oracle/m3t_oracle.h", "m3t_oracle_")
    lib = ctypes.CDLL(util.build_oracle())
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    # both sides expose the same tracker / modality surface
    hip = {n[len("m3t_hip_"):] for n in _declared("include/m3t_hip.h", "m3t_hip_")}
    # (execute_tracking_step_parallel: the OpenMP variant bench.py times as the all-cores CPU baseline, oracle only)
    ora = {n[len("m3t_oracle_"):] for n in names if ("histograms_" not in n or "modality" in n)} - {"execute_tracking_step_parallel"}
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


def test_python_structs_match_the_c_structs_field_for_field(tmp_path):
    """the ctypes mirrors in _capi.py against include/m3t_types.h as a C compiler lays it out: same size, same
    field names, same offsets (a drifted field would shift every parameter behind it silently)"""
    import subprocess
    capi = util.pkg._capi
    pairs = [("m3t_intrinsics", capi.Intrinsics), ("m3t_region_model_desc", capi.RegionModelDesc),
             ("m3t_depth_model_desc", capi.DepthModelDesc), ("m3t_region_modality_params", capi.RegionModalityParams),
             ("m3t_depth_modality_params", capi.DepthModalityParams), ("m3t_body_geometry", capi.BodyGeometry),
             ("m3t_model_generation_params", capi.ModelGenerationParams), ("m3t_data_line", capi.DataLine),
             ("m3t_data_point", capi.DataPoint)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "m3t_types.h"', "int main(void) {"]
    for c_name, py in pairs:
        lines.append('  printf("%s %%zu\\n", sizeof(%s));' % (c_name, c_name))
        for field, _ in py._fields_:
            lines.append('  printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (c_name, field, c_name, field))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o",
                           str(exe)])
    layout = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    header = open(os.path.join(ROOT, "include", "m3t_types.h")).read()
    for c_name, py in pairs:
        assert int(layout[c_name]) == ctypes.sizeof(py), c_name
        for field, _ in py._fields_:
            assert int(layout[c_name + "." + field]) == getattr(py, field).offset, (c_name, field)
        # and no C field without a Python counterpart: the field counts agree
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (c_name, c_name), header, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        n_c_fields = sum(d.count(",") + 1 for d in body.split(";") if d.strip())  # 'float fu, fv;' declares two
        assert n_c_fields == len(py._fields_), (c_name, n_c_fields, len(py._fields_))
    # the numpy record types the getters fill are the same records
    assert capi.DATA_LINE_DTYPE.itemsize == ctypes.sizeof(capi.DataLine)
    assert capi.DATA_POINT_DTYPE.itemsize == ctypes.sizeof(capi.DataPoint)


def test_python_signatures_have_the_declared_argument_counts():
    """every binding in _capi.py takes as many arguments as its prototype in include/m3t_hip.h (context first)"""
    capi = util.pkg._capi
    txt = open(os.path.join(ROOT, "include", "m3t_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    prototypes = {m.group(1): m.group(2) for m in re.finditer(r"\bint\s+m3t_hip_(\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S)}
    bindings = dict(capi._SIGNATURES)
    bindings.update(capi._HIP_ONLY)
    checked = 0
    for name, args in bindings.items():
        assert name in prototypes, name
        declared = [a for a in prototypes[name].split(",") if a.strip() and a.strip() != "void"]
        assert len(declared) == len(args) + 1, (name, prototypes[name], args)
        checked += 1
    assert checked >= 70
    unbound = set(prototypes) - set(bindings) - {"create", "destroy", "last_error"}
    assert not unbound, sorted(unbound)  # the Python mirror reaches the whole boundary


def test_cpp_mirror_reaches_the_whole_boundary():
    """every entry point of include/m3t_hip.h is wrapped by the C++ mirror headers"""
    names = _declared("include/m3t_hip.h", "m3t_hip_")
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("m3t_hip.hpp", "m3t_hip_config.hpp"))
    assert [n for n in names if n not in text] == []
