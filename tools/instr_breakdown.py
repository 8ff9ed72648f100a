"""Developer tool: which part of a frame the tracking kernel's instructions belong to.  Runs the fused step of 64
RBOT objects with growing iteration counts (n_corr, n_update); under
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES
the differences between consecutive settings are the per-phase instruction counts (tools/instr_breakdown_summary.py).
Launch shape through M3T_HIP_NO_SPLIT / M3T_HIP_THREADS."""
import ctypes as C, importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("3dobjecttracking_amd")
import scenes
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "rbot"  # rbot | ycb | ycb_region | ycb_region_noocc | ycb_depth
ycb = mode != "rbot"
SERIES = [(1, 0), (1, 1), (1, 2), (2, 2), (3, 2), (4, 2), (5, 2), (7, 2)]
hip = pkg.open_context(0)
inputs = scenes.Inputs(n_obj, 2 + 3 * len(SERIES), n_divides=4, n_models=8, with_depth=ycb)
import importlib
syn = importlib.import_module("3dobjecttracking_amd.synthetic")
region_params = dict(syn.YCB_REGION_PARAMS, measure_occlusions=0) if mode == "ycb_region_noocc" else None
inst = scenes.Instance(hip, inputs, region_params=region_params, use_region=mode != "ycb_depth",
                       use_depth=mode in ("ycb", "ycb_depth"))
inst.upload_frame(0)
inst.tracker.StartModalities(0)
k = 1
for n_corr, n_update in SERIES:
    hip.call("tracker_set_iterations", n_corr, n_update)
    for _ in range(3):
        inst.upload_frame(k)
        inst.tracker.ExecuteTrackingStep(k)
        k += 1
    hip.call("sync")
    shape = (C.c_int * 4)()
    hip.call("get_step_shape", shape)
    print("series", n_corr, n_update, list(shape), flush=True)
