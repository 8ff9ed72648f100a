"""Developer tool: s_memtime cycles of the structure phases of tracking_step_tree_kernel (-DM3T_PHASE_TIMING build)."""
import ctypes as C, importlib, os, sys

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")  # inputs on worker processes (same bits; batch.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("3dobjecttracking_amd")
import os
import bench_chain
import bench_inputs as scenes
lib = sys.argv[1]
hip = pkg.CApi(lib, "m3t_hip_")
f = hip.lib.m3t_hip_debug_phase_cycles
f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
syn, host = pkg.synthetic, pkg.host
inputs, joints, gt = bench_chain.chain_inputs(scenes, syn, 8, 8, 2)
start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
ch = bench_chain.Chain(hip, host, syn, inputs, joints, start_root, gt[0][1] + 0.01, range(8))
ch.upload(inputs, 0); ch.tracker.StartModalities(0)
buf = (C.c_ulonglong * 32)()
for k in range(1, 4):
    ch.upload(inputs, k); ch.tracker.ExecuteTrackingStep(k)
f(hip.ctx, buf, 1)
n = 4
for k in range(4, 8):
    ch.upload(inputs, k); ch.tracker.ExecuteTrackingStep(k)
f(hip.ctx, buf, 1)
NAMES = {
    0: "search: view", 1: "search: phase A (lines)", 2: "search: phase B (pixels)", 3: "search: phase C1 (distributions)",
    4: "search: phase C2 (normalisation)",
    7: "  B: addresses + pixel load issue", 8: "  B: pixel wait", 9: "  B: gather issue", 10: "  B: gather wait",
    11: "  B: products",
    5: "Newton: g/H products + barrier (all waves; last wave: adjoints)",
    17: "  last wave: adjoints of all links", 18: "  last wave: Jacobians, a column per lane",
    23: "Newton: g/H chain (42 lanes) + publish", 22: "Newton: collect the other links' sums + barrier",
    30: "Newton: system -sum J^T H J | sum J^T g (wave per link)",
    31: "Newton: solve + pose updates (first wave) + barrier",
    12: "  solve: constraint rows", 13: "  solve: LDL^T", 14: "  solve: exp() of the variations",
    15: "  solve: link2world down the tree",
    19: "  system: H J + the link's terms (wave per link)", 20: "  system: barrier (the slowest wave)",
    21: "  solve: joints (sixteen lanes per link)",
    26: "histogram update (tail)", 27: "  tail: view", 28: "  tail: occlusion windows", 29: "  tail: pixel walk",
}
ORDER = [0, 1, 2, 7, 8, 9, 10, 11, 3, 4, 5, 17, 18, 23, 22, 30, 19, 20, 31, 12, 13, 14, 21, 15, 26, 27, 28, 29]
name, shape = C.create_string_buffer(64), (C.c_int * 4)()
hip.call("get_step_kernel", name, 64)
hip.call("get_step_shape", shape)
print("%s %s, 8-body chain (13 dof), s_memtime ticks per frame of workgroup 0 (7 searches, 14 Newton steps)" % (name.value.decode(), list(shape)))
top = 0
for i in ORDER + [j for j in range(32) if j not in ORDER]:
    if buf[i]:
        name = NAMES.get(i, "phase %d" % i)
        print("%-66s %9.0f" % (name, buf[i] / n))
        if not name.startswith("  "):
            top += buf[i] / n
print("%-66s %9.0f" % ("sum of the top-level phases", top))
