"""Developer tool: s_memtime cycles of the structure phases of tracking_step_tree_kernel (-DM3T_PHASE_TIMING build)."""
import ctypes as C, importlib, os, sys

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")  # inputs on worker processes (same bits; batch.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("3dobjecttracking_amd")
import os
import bench_chain
scenes = pkg.batch
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
for i in range(32):
    if buf[i]:
        print("phase %2d: %9.0f cycles/frame" % (i, buf[i] / n))
