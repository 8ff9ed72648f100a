"""One worker of bench.py's all-cores CPU leg: tracks `count` objects of the bench workload with the
oracle (one thread) for `seconds` and prints {"pose_updates": n, "seconds": t}.  The start of the timed
loop is aligned across workers through a shared start time.
usage: cpu_baseline_worker.py first_object count n_frames n_divides start_epoch seconds"""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import util  # noqa: E402

first, count, n_frames, n_divides = (int(x) for x in sys.argv[1:5])
start_epoch, seconds = float(sys.argv[5]), float(sys.argv[6])
inputs = scenes.Inputs(count, n_frames, n_divides=n_divides, n_models=1, first_object=first)
ora = util.open_oracle()
inst = scenes.Instance(ora, inputs)
inst.upload_frame(0)
inst.tracker.StartModalities(0)
ready = time.time()
while time.time() < start_epoch:
    time.sleep(0.005)
done, spent = 0, 0.0
t_begin = time.time()
while time.time() - t_begin < seconds:
    for k in range(1, n_frames):
        inst.upload_frame(k)
        t = time.perf_counter()
        inst.tracker.ExecuteTrackingStep(k)
        spent += time.perf_counter() - t
        done += count
        if time.time() - t_begin >= seconds:
            break
    inst.set_poses([inputs.gt[i][0] for i in range(count)])
print(json.dumps({"pose_updates": done, "seconds": spent, "late_s": max(0.0, ready - start_epoch)}))
