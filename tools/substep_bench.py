"""Fused kernel vs one launch per sub-step (the path kinematic structures and renderer-fed
configurations take): wall time per tracking step.  python tools/substep_bench.py [n_objects]"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401

torch.cuda.init()
import scenes  # noqa: E402
import util  # noqa: E402

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_frames = 12
inputs = scenes.Inputs(n_obj, n_frames, n_divides=2, n_models=4)
for mode in (1, 0):
    hip = util.open_hip()
    hip.call("set_fused_step", mode)
    inst = scenes.Instance(hip, inputs)
    for cam in inst.color_cams:
        cam.set_ring(n_frames)
    for k in range(n_frames):
        for i, cam in enumerate(inst.color_cams):
            cam.upload_slot(k, inputs.color[i][k])
    inst.tracker.select_slot(0)
    inst.tracker.StartModalities(0)
    for k in range(1, 4):
        inst.tracker.select_slot(k)
        inst.tracker.ExecuteTrackingStep(k)
    hip.call("sync")
    t0 = time.perf_counter()
    reps = 5
    for r in range(reps):
        for k in range(4, n_frames):
            inst.tracker.select_slot(k)
            inst.tracker.ExecuteTrackingStep(k)
    hip.call("sync")
    el = (time.perf_counter() - t0) / (reps * (n_frames - 4))
    print(f"fused_step={mode}: {el * 1e3:.3f} ms per step, {n_obj / el:.0f} pose-updates/s")
