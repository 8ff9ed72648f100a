"""Developer tool: the ROI ingest loop of bench.py's pcie leg on its own (64 RBOT objects, rectangles pulled from one
page-locked block per batch-frame while the previous step runs), for a kernel trace:

  rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/roi_trace.py     then
  python tools/roi_trace.py --report DIR    prints when the tracking kernel and the pull kernel of the last rounds ran

Without rocprofv3 it prints the loop's ms per step.  ROI_RESERVE_CUS=32: with m3t_hip_reserve_ingest_cus(32).  ROI_MARGIN: pixels."""
import csv
import ctypes as C
import glob
import importlib
import os
import sys
import time

import numpy as np

os.environ.setdefault("M3T_INPUT_WORKERS", "auto")  # inputs on worker processes (same bits; bench_inputs.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def report(directory):
    files = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    keep = [r for r in rows if any(k in r[2] for k in ("tracking_step", "roi_pull", "roi_rect", "roi_check"))]
    keep = keep[-24:]
    t0 = keep[0][0]
    for s, e, name in keep:
        print("%-34s start %9.1f us  end %9.1f us  (%.1f us)" % (name.split("(")[0][:34], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    pkg = importlib.import_module("3dobjecttracking_amd")
    import bench_inputs as scenes
    n_obj, n_frames = 64, 8
    margin = float(os.environ.get("ROI_MARGIN", "24"))
    hip = pkg.open_context(0)
    if os.environ.get("ROI_RESERVE_CUS"):  # the pull kernel on CUs of its own (m3t_hip_reserve_ingest_cus)
        hip.call("reserve_ingest_cus", int(os.environ["ROI_RESERVE_CUS"]))
    inputs = scenes.Inputs(n_obj, n_frames, n_divides=2, n_models=8)
    inst = scenes.Instance(hip, inputs)
    blocks = [np.stack([inputs.color[i][k] for i in range(n_obj)]) for k in range(n_frames)]
    for b in blocks:
        inst.tracker.register_host_buffer(b)
    ids = (C.c_int * n_obj)(*[cam.id for cam in inst.color_cams])
    hip.call("cameras_set_ring", ids, n_obj, 2)
    hip.call("set_roi_ingest", 1, C.c_float(margin))

    def upload(fn, slot, b):
        hip.call(fn, ids, n_obj, slot, b.ctypes.data_as(C.c_void_p), b.strides[0], b.strides[1])

    upload("cameras_upload_batch_async", 0, blocks[0])
    hip.call("ingest_sync")
    hip.call("cameras_select_slot", 0)
    hip.call("start_modalities", 0)
    upload("cameras_upload_batch_async", 1, blocks[1])
    hip.call("ingest_sync")
    hip.call("sync")
    for rep in range(3):
        t0 = time.perf_counter()
        n = 0
        for k in range(1, n_frames - 1):
            hip.call("cameras_select_slot", k % 2)
            hip.call("execute_tracking_step", k)
            upload("cameras_upload_batch_roi_async", (k + 1) % 2, blocks[k + 1])
            n += 1
        hip.call("ingest_sync")
        hip.call("sync")
        dt = time.perf_counter() - t0
        # (restart: whole frame 1 into slot 1, poses as they are -- the loop only has to be representative)
        upload("cameras_upload_batch_async", 1, blocks[1])
        hip.call("ingest_sync")
        hip.call("sync")
    bodies = (C.c_int * 64)()
    n_miss, pulls = C.c_int(0), C.c_longlong(0)
    hip.call("roi_get_status", bodies, 64, C.byref(n_miss), C.byref(pulls))
    print("roi loop: %.3f ms per step (%d steps), %d rectangle uploads, %d bodies outside" % (dt / n * 1e3, n, pulls.value, n_miss.value))


if __name__ == "__main__":
    main()
