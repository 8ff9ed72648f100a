"""PCIe ingest micro-benchmark: page-locked frames through m3t_hip_camera_upload_slot_async with and
without tracking steps in between (run on the GPU box: python tools/ingest_bench.py)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (initialise torch's HIP runtime first, as tests/conftest.py does)

torch.cuda.init()
import scenes  # noqa: E402
import util  # noqa: E402

n_obj, n_frames = 64, 8
inputs = scenes.Inputs(n_obj, n_frames, n_divides=2, n_models=4)
hip = util.open_hip()
inst = scenes.Instance(hip, inputs)
inst.upload_frame(0)
inst.tracker.StartModalities(0)
blocks = [np.stack([inputs.color[i][k] for i in range(n_obj)]) for k in range(n_frames)]
for b in blocks:
    inst.tracker.register_host_buffer(b)
for cam in inst.color_cams:
    cam.set_ring(2)
nbytes = blocks[0].nbytes
for with_steps in (False, True):
    for i, cam in enumerate(inst.color_cams):
        cam.upload_slot(1, blocks[1][i], asynchronous=True)
    inst.tracker.ingest_sync()
    hip.call("sync")
    t0 = time.perf_counter()
    reps = 3
    for r in range(reps):
        for k in range(1, n_frames):
            if with_steps:
                inst.tracker.select_slot(k % 2)
                inst.tracker.ExecuteTrackingStep(k)
            nk = k + 1 if k + 1 < n_frames else 1
            for i, cam in enumerate(inst.color_cams):
                cam.upload_slot((k + 1) % 2, blocks[nk][i], asynchronous=True)
    t_enq = time.perf_counter() - t0
    inst.tracker.ingest_sync()
    hip.call("sync")
    el = time.perf_counter() - t0
    n = reps * (n_frames - 1)
    print(f"with_steps={with_steps}: {el / n * 1e3:.3f} ms per batch of {n_obj} frames, "
          f"{nbytes * n / el / 1e9:.1f} GB/s, enqueue {t_enq / n * 1e3:.3f} ms")

# what the runtime gives a plain page-locked tensor copy of the same size (upper bound for this box)
h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"torch pinned copy of one {nbytes >> 20} MiB block: {nbytes * 10 / el / 1e9:.1f} GB/s")
