"""examples/run_generated_tracker.cpp over the device context: build the tracker a YAML configuration
describes (generator.h), detect, start, and track the loader cameras' image sequence.

    python tools/run_generated_tracker.py CONFIG.yaml [N_FRAMES]

Prints one line per tracked frame: frame index, wall time of the step, and the bodies' poses (row-major 3x4)."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3dobjecttracking_amd")


def main(argv):
    if len(argv) < 2:
        sys.stderr.write("Not enough arguments: Provide configfile_path\n")
        return -1
    n_frames = int(argv[2]) if len(argv) > 2 else 1 << 30
    api = pkg.open_context(0)
    tracker = pkg.generator.GenerateConfiguredTracker(api, argv[1])
    if not tracker.SetUp():
        return -1
    names = {o.name for o in tracker.optimizers}
    if not (tracker.DetectPoses(names) and tracker.StartModalities(0)):
        return -1
    for iteration in range(n_frames):
        if iteration > 0 and not tracker.UpdateCameras(iteration):
            break
        t0 = time.perf_counter()
        if not (tracker.ExecuteTrackingStep(iteration) and tracker.Sync()):
            return -1
        ms = (time.perf_counter() - t0) * 1e3
        poses = "  ".join("%s: %s" % (b.name, np.array2string(b.body2world_pose()[:3].reshape(-1), precision=5,
                                                              max_line_width=1000)) for b in tracker.body_ptrs())
        print("frame %d  %.3f ms  %s" % (iteration, ms, poses))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
