"""Scene builders shared by the GPU parity tests, __graft_entry__.smoke() and bench.py:
the same seeded synthetic inputs are instantiated behind two C-ABI contexts (the HIP
product and the CPU oracle) and driven through identical Tracker calls."""
import numpy as np

import util
from util import host, syn


# the batch builders live beside bench.py (bench_inputs.py: bench.py uses them without touching tests/)
import bench_inputs  # noqa: E402 (repo root: bench / test infrastructure, not in the product package)

Inputs = bench_inputs.Inputs
Instance = bench_inputs.Instance
replicate = bench_inputs.replicate
subset = bench_inputs.subset
stage_frames = bench_inputs.stage_frames


def compare_poses(pa, pb):
    rot = max(syn.pose_errors(a, b)[0] for a, b in zip(pa, pb))
    trans = max(syn.pose_errors(a, b)[1] for a, b in zip(pa, pb))
    return rot, trans


def run_region_parity(hip, ora, n_objects=2, n_frames=2, n_divides=2, resync=True, verbose=False, inputs=None,
                      **kw):
    """StartModalities on frame 0 then ExecuteTrackingStep per frame on both sides.
    resync=True: single-step parity (device pose + histograms re-synchronised from the
    oracle before every frame); False: free running (ii).
    Returns (max rotation diff [rad], max translation diff [m]) over objects and frames."""
    inputs = inputs or Inputs(n_objects, n_frames, n_divides=n_divides)
    a = Instance(hip, inputs, **kw)
    b = Instance(ora, inputs, **kw)
    a.upload_frame(0)
    b.upload_frame(0)
    assert a.tracker.StartModalities(0) and b.tracker.StartModalities(0)
    worst = (0.0, 0.0)
    for k in range(inputs.n_frames):
        a.upload_frame(k)
        b.upload_frame(k)
        if resync:
            a.set_poses(b.poses())
            for ra, rb in zip(a.region, b.region):
                ra.set_histograms(*rb.histograms())
        assert a.tracker.ExecuteTrackingStep(k) and b.tracker.ExecuteTrackingStep(k)
        rot, trans = compare_poses(a.poses(), b.poses())
        worst = (max(worst[0], rot), max(worst[1], trans))
        if verbose:
            gt_err = [syn.pose_errors(p, inputs.gt[i][k]) for i, p in enumerate(b.poses())]
            print("frame %d: hip-vs-oracle rot %.3g rad trans %.3g m | oracle-vs-gt rot %.3g trans %.3g" %
                  (k, rot, trans, max(e[0] for e in gt_err), max(e[1] for e in gt_err)))
    return worst
