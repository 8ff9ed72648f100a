"""Developer tool: pose-updates/s of several builds of the library and several batch sizes in ONE process (the
rendered input streams are built once).  Not the judged benchmark (that is bench.py); used to compare kernel variants.

  python tools/quick_bench.py [--ycb] [--objects 64,4096] [--steps 20] [--env K=V,K=V] lib_a.so [lib_b.so ...]

Prints one line per (library, batch): ms/step (median of 3 timed regions), pose-updates/s, launch shape, and the mean
duration of the tracking kernel and of the separate histogram kernel by HIP events."""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")  # inputs on worker processes (same bits; bench_inputs.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dobjecttracking_amd")
import bench_inputs as scenes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ycb", action="store_true")
    ap.add_argument("--objects", default="64")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--models", type=int, default=8)
    ap.add_argument("--env", default="", help="semicolon separated environment sets, each K=V,K=V (one run per set)")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    counts = [int(x) for x in a.objects.split(",")]
    K, W = a.steps, a.warmup
    n_frames = min(K + W + 1, 12)  # frames are cycled (the ring holds n_frames)
    base = scenes.Inputs(min(a.streams, max(counts)), n_frames, n_divides=4, n_models=min(a.models, max(counts)),
                         with_depth=a.ycb)
    env_sets = [e for e in a.env.split(";")] if a.env else [""]
    for lib in a.libs:
        for n in counts:
            for env in env_sets:
                sets = dict(kv.split("=") for kv in env.split(",") if kv)
                for k in [k for k in os.environ if k.startswith("M3T_HIP_")]:
                    os.environ.pop(k, None)
                os.environ.update(sets)
                hip = pkg.CApi(lib, "m3t_hip_")
                rep = scenes.replicate(base, n)
                inst = scenes.Instance(hip, rep, use_depth=a.ycb)
                scenes.stage_frames(hip, inst, rep, n_frames)
                hip.call("cameras_select_slot", 0)
                hip.call("start_modalities", 0)

                def run(count, first=1):
                    for k in range(first, first + count):
                        hip.call("cameras_select_slot", 1 + (k % (n_frames - 1)))
                        hip.call("execute_tracking_step", k)

                run(W)
                hip.call("sync")
                times = []
                for _ in range(3):
                    t = time.perf_counter()
                    run(K, W + 1)
                    hip.call("sync")
                    times.append(time.perf_counter() - t)
                el = float(np.median(times))
                hip.call("set_kernel_timing", 1)
                run(K, W + 1)
                ms = (C.c_float * 2)()
                cnt = (C.c_int * 2)()
                hip.call("get_kernel_timing", ms, cnt)
                hip.call("set_kernel_timing", 0)
                shape = (C.c_int * 4)()
                hip.call("get_step_shape", shape)
                kernel = ""
                if "get_step_kernel" in hip._fn:
                    nb = C.create_string_buffer(64)
                    hip.call("get_step_kernel", nb, 64)
                    kernel = nb.value.decode()
                poses = np.zeros((n, 16), np.float32)
                hip.call("bodies_get_poses", poses.ctypes.data_as(C.POINTER(C.c_float)), n)
                print("%-28s %-26s n=%5d  %.4f ms/step  %9.0f pose-updates/s  shape %s  track %.4f ms  hist %.4f ms  "
                      "pose-sum %.9g %s" % (os.path.basename(os.path.dirname(lib)) + "/" + os.path.basename(lib), env, n,
                                            el / K * 1e3, n * K / el, list(shape), ms[0] / max(cnt[0], 1),
                                            ms[1] / max(cnt[1], 1), float(np.abs(poses[:8]).sum()), kernel), flush=True)
                del inst, hip


if __name__ == "__main__":
    main()
