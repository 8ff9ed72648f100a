"""Developer tool: ms per tracking step for several launch shapes on the same inputs (built once), with a check
that every shape ends on the same poses bit for bit.  usage: sweep_shapes.py [rbot|ycb] ..."""
import ctypes as C, importlib, json, os, sys, time

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")  # inputs on worker processes (same bits; batch.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
pkg = importlib.import_module("3dobjecttracking_amd")
import os
import scenes


def subset(inputs, n):
    sub = scenes.Inputs.__new__(scenes.Inputs)
    sub.__dict__.update(inputs.__dict__)
    sub.n_objects = n
    return sub


def measure(inputs, use_depth, env, K=8, W=3, repeats=3):
    for k in ("M3T_HIP_NO_SPLIT", "M3T_HIP_SPLIT_PARTS", "M3T_HIP_THREADS", "M3T_HIP_NO_FUSED_HISTOGRAM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    hip = pkg.open_context(0)
    inst = scenes.Instance(hip, inputs, use_depth=use_depth)
    n_frames = inputs.n_frames
    for cams, frames in ((inst.color_cams, inputs.color), (inst.depth_cams, inputs.depth)):
        for i, cam in enumerate(cams):
            if cam is None:
                continue
            hip.call("camera_set_ring", cam.id, n_frames)
            for k in range(n_frames):
                f = frames[i][k]
                hip.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
    hip.call("cameras_select_slot", 0)
    hip.call("start_modalities", 0)
    best = []
    for r in range(repeats):
        inst.set_poses(inputs.start)
        for k in range(1, 1 + W):
            hip.call("cameras_select_slot", k); hip.call("execute_tracking_step", k)
        hip.call("sync")
        t = time.perf_counter()
        for k in range(1 + W, 1 + W + K):
            hip.call("cameras_select_slot", k); hip.call("execute_tracking_step", k)
        hip.call("sync")
        best.append((time.perf_counter() - t) / K * 1e3)
    shape = (C.c_int * 4)()
    hip.call("get_step_shape", shape)
    poses = np.zeros((inputs.n_objects, 16), np.float32)
    hip.call("bodies_get_poses", poses.ctypes.data_as(C.POINTER(C.c_float)), inputs.n_objects)
    for k in env:
        os.environ.pop(k, None)
    return min(best), float(np.median(best)), list(shape), poses


def main():
    what = sys.argv[1:] or ["rbot", "ycb"]
    if "rbot64" in what:
        inputs = scenes.Inputs(64, 12, n_divides=4, n_models=8)
        for env in ({"M3T_HIP_NO_SPLIT": "1"}, {"M3T_HIP_SPLIT_PARTS": "4"}):
            mn, med, shape, poses = measure(inputs, False, env)
            print(json.dumps({"config": "rbot", "objects": 64, "env": env, "shape": shape, "ms_min": round(mn, 4),
                              "ms_median": round(med, 4), "k_pose_updates_s": round(64 / mn, 1)}), flush=True)
    if "rbot" in what:
        inputs = scenes.Inputs(64, 12, n_divides=4, n_models=8)
        for n, envs in ((64, [{"M3T_HIP_NO_SPLIT": "1"}, {"M3T_HIP_SPLIT_PARTS": "2"}, {"M3T_HIP_SPLIT_PARTS": "4"},
                              {"M3T_HIP_SPLIT_PARTS": "8", "M3T_HIP_THREADS": "256"},
                              {"M3T_HIP_SPLIT_PARTS": "4", "M3T_HIP_THREADS": "256"}]),
                        (32, [{"M3T_HIP_SPLIT_PARTS": "4"}, {"M3T_HIP_SPLIT_PARTS": "8"}]),
                        (16, [{"M3T_HIP_SPLIT_PARTS": "8"}, {"M3T_HIP_SPLIT_PARTS": "16"}]),
                        (1, [{"M3T_HIP_NO_SPLIT": "1"}, {"M3T_HIP_SPLIT_PARTS": "8"}, {"M3T_HIP_SPLIT_PARTS": "16"}])):
            ref = None
            for env in envs:
                mn, med, shape, poses = measure(subset(inputs, n), False, env)
                same = ref is None or bool(np.array_equal(ref, poses))
                ref = poses if ref is None else ref
                print(json.dumps({"config": "rbot", "objects": n, "env": env, "shape": shape, "ms_min": round(mn, 4),
                                  "ms_median": round(med, 4), "k_pose_updates_s": round(n / mn, 1), "same_bits": same}),
                      flush=True)
    if "ycb" in what:
        inputs = scenes.Inputs(21, 12, n_divides=4, n_models=6, with_depth=True)
        ref = None
        for env in ({"M3T_HIP_NO_SPLIT": "1"}, {"M3T_HIP_SPLIT_PARTS": "8"}):
            mn, med, shape, poses = measure(inputs, True, env)
            same = ref is None or bool(np.array_equal(ref, poses))
            ref = poses if ref is None else ref
            print(json.dumps({"config": "ycb", "objects": 21, "env": env, "shape": shape, "ms_min": round(mn, 4),
                              "ms_median": round(med, 4), "k_pose_updates_s": round(21 / mn, 1), "same_bits": same}),
                  flush=True)


if __name__ == "__main__":
    main()
