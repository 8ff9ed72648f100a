"""Developer tool: per-phase s_memtime cycles of the fused tracking kernel (block 0),
using a -DM3T_PHASE_TIMING build of the library (gpurun_out/libm3t_hip_timing.so)."""
import ctypes as C, importlib, os, sys

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")  # inputs on worker processes (same bits; bench_inputs.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("3dobjecttracking_amd")
import os
import scenes
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "libm3t_hip_timing.so")
n_obj = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ycb = len(sys.argv) > 3 and sys.argv[3] == "ycb"
hip = pkg.CApi(lib, "m3t_hip_")
f = hip.lib.m3t_hip_debug_phase_cycles
f.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
inputs = scenes.Inputs(n_obj, 8, n_divides=4, n_models=min(8, n_obj), with_depth=ycb)
inst = scenes.Instance(hip, inputs, use_depth=ycb)
inst.upload_frame(0)
inst.tracker.StartModalities(0)
buf = (C.c_ulonglong * 32)()
for k in range(1, 4):
    inst.upload_frame(k); inst.tracker.ExecuteTrackingStep(k)
f(hip.ctx, buf, 1)
n = 4
for k in range(4, 8):
    inst.upload_frame(k); inst.tracker.ExecuteTrackingStep(k)
f(hip.ctx, buf, 1)
names = ["view search", "phase A (lines)", "phase B (pixels)", "phase C1 (dist)", "phase C2 (moments)",
         "g/H products + barrier (u=0, global)", "solve (wave) + barrier",
         "  B: addr+issue", "  B: pixel wait", "  B: gather issue", "  B: gather wait", "  B: products",
         "  solve: permute + gather", "  solve: LDLT", "  solve: trisolve", "  solve: expm", "depth scan (all)",
         "  d: view search", "  d: point setup", "  d: window scan", "  d: reduce+occlusion", "  d: write",
         "split exchange", "g/H chain (42 lanes)", "g/H products + barrier (u>=1, local)", "moments / depth vote",
         "histogram update (tail)"]
tot = sum(buf[i] for i in range(7)) + buf[16] + sum(buf[22:27])
for i, nme in enumerate(names):
    print("%-38s %10.0f cycles/frame  %5.1f%%" % (nme, buf[i] / n, 100.0 * buf[i] / tot))
print("total %.0f cycles/frame" % (tot / n))
print("tail: view search %.0f, occlusion windows %.0f, pixel walk %.0f cycles/frame (blend = the rest)" %
      (buf[27] / n, buf[28] / n, buf[29] / n))

# split exchange of object 0, last frame: when each part published and when it had everybody's results
try:
    g = hip.lib.m3t_hip_debug_exchange_times
    g.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    xt = (C.c_ulonglong * 768)()
    g(hip.ctx, xt)
    shape = (C.c_int * 4)()
    hip.call("get_step_shape", shape)
    parts = shape[1]
    if parts > 1:
        print("split exchange, object 0 (cycles relative to the first part's publish start of the round):")
        for rnd in range(16):
            start = [xt[(0 * 16 + rnd) * 16 + p] for p in range(parts)]
            if not any(start):
                break
            pub = [xt[(1 * 16 + rnd) * 16 + p] for p in range(parts)]
            done = [xt[(2 * 16 + rnd) * 16 + p] for p in range(parts)]
            t0 = min(start)
            print("  round %d: publish start %s  published %s  collected %s" %
                  (rnd, [int(v - t0) for v in start], [int(v - t0) for v in pub], [int(v - t0) for v in done]))
except AttributeError:
    pass
