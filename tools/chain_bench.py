"""Developer tool: ms per tracking step of the 8-body chain (bench.py --config chain8's structure) for several builds
of the library in ONE process on the same frames: the one-launch step, and with --distributed the path a rank runs
when the structure spans GPUs (library communicator at world size 1).  Prints the kernel, a pose checksum (the bits
must agree between builds) and, with --oracle, whether the poses equal the CPU oracle's.

  python tools/chain_bench.py [--bodies 8] [--steps 20] [--distributed] [--oracle] lib_a.so [lib_b.so ...]"""
import argparse
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
pkg = importlib.import_module("3dobjecttracking_amd")
import bench_chain  # noqa: E402

import bench_inputs as scenes  # noqa: E402

syn, host = pkg.synthetic, pkg.host


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bodies", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--regions", type=int, default=7)
    ap.add_argument("--distributed", action="store_true")
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    K, W, n = a.steps, a.warmup, a.bodies
    n_frames = K + W + 1
    inputs, joints, gt = bench_chain.chain_inputs(scenes, syn, n, n_frames, 2)
    start_root = syn.perturb_pose(gt[0][0][0], np.random.default_rng(5), rot_deg=0.5, trans=0.001)
    start_angles = gt[0][1] + 0.01
    reference = None
    if a.oracle:
        import util
        ora = util.open_oracle()
        oc = bench_chain.Chain(ora, host, syn, inputs, joints, start_root, start_angles, range(n))
        oc.upload(inputs, 0)
        assert oc.tracker.StartModalities(0)
        for k in range(1, 1 + W + K):
            oc.upload(inputs, k)
            assert oc.tracker.ExecuteTrackingStep(k)
        reference = oc.poses()
    for lib in a.libs:
        for distributed in ([False, True] if a.distributed else [False]):
            hip = pkg.CApi(lib, "m3t_hip_")
            ch = bench_chain.Chain(hip, host, syn, inputs, joints, start_root, start_angles, range(n))
            if distributed:
                buf = C.create_string_buffer(128)
                hip.call("comm_get_unique_id", buf, 128)
                hip.call("comm_init_rank", buf, 128, 1, 0)
            for i, cam in enumerate(ch.cams):
                hip.call("camera_set_ring", cam.id, n_frames)
                for k in range(n_frames):
                    f = inputs.color[i][k]
                    hip.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])

            def run(first, count):
                for k in range(first, first + count):
                    hip.call("cameras_select_slot", k)
                    hip.call("execute_tracking_step", k)

            hip.call("cameras_select_slot", 0)
            hip.call("start_modalities", 0)
            run(1, W)
            hip.call("sync")
            times, first = [], None
            for _ in range(a.regions):
                t = time.perf_counter()
                run(1 + W, K)
                hip.call("sync")
                times.append(time.perf_counter() - t)
                if first is None:
                    first = ch.poses()
            name = C.create_string_buffer(64)
            hip.call("get_step_kernel", name, 64)
            shape = (C.c_int * 4)()
            hip.call("get_step_shape", shape)
            eq = "" if reference is None else ("  == oracle: %s" % bool(np.array_equal(first, reference)))
            print("%-34s %-12s %.4f ms/step (min %.4f)  %8.0f pose-updates/s  %s %s  pose-sum %.9g%s" % (
                os.path.basename(os.path.dirname(lib)) + "/" + os.path.basename(lib),
                "distributed" if distributed else "one launch", float(np.median(times)) / K * 1e3, min(times) / K * 1e3,
                n * K / float(np.median(times)), name.value.decode() or "(sub-step launches)", list(shape),
                float(np.abs(first).sum()), eq), flush=True)
            if distributed:
                hip.call("comm_destroy")
            del ch, hip


if __name__ == "__main__":
    main()
