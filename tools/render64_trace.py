"""Developer tool: the renderer-fed step of the reference's test scene 64 times in one context (bench.py --extras,
renderer_fed_64_objects) by itself, for rocprofv3 --kernel-trace --stats:
  rocprofv3 --kernel-trace --stats --output-format csv -d out -- python tools/render64_trace.py [n_objects] [steps]"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
pkg = importlib.import_module("3dobjecttracking_amd")
import golden_scene as gs
from util import host

n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
api = pkg.CApi(os.environ["M3T_LIB"], "m3t_hip_") if os.environ.get("M3T_LIB") else pkg.open_context(0)  # (M3T_LIB: a variant build)
fixtures = []
for _ in range(n_obj):
    g = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                          depth_params=dict(n_unoccluded_iterations=0))
    geo, _ = gs.fixture_renderer_geometry(api, g.body)
    rs = (host.FocusedBasicDepthRenderer(api, geo, g.color_camera),
          host.FocusedSilhouetteRenderer(api, geo, g.color_camera, id_type=1),
          host.FocusedBasicDepthRenderer(api, geo, g.depth_camera),
          host.FocusedSilhouetteRenderer(api, geo, g.depth_camera, id_type=0))
    for r in rs:
        r.AddReferencedBody(g.body)
    g.region.ModelOcclusions(rs[0])
    g.region.UseRegionChecking(rs[1])
    g.depth.ModelOcclusions(rs[2])
    g.depth.UseSilhouetteChecking(rs[3])
    fixtures.append((g, rs))
start = fixtures[0][0].body.body2world_pose()
tracker = fixtures[0][0].tracker
import ctypes as C
start_all = (C.c_float * (16 * 2 * n_obj))()  # every fixture made two bodies: the tracked triangle and the bottle
api.call("bodies_get_poses", start_all, 2 * n_obj)
tracker.StartModalities(0)
tracker.ExecuteTrackingStep(0)
api.call("sync")
poses = np.stack([g.body.body2world_pose() for g, _ in fixtures])
t0 = time.perf_counter()
for _ in range(steps):
    api.call("bodies_set_poses", start_all, 2 * n_obj)  # (one call: 2 n set_body2world_pose calls are not the tracker's time)
    tracker.ExecuteTrackingStep(0)
api.call("sync")
dt = (time.perf_counter() - t0) / steps
print("%d objects: %.3f ms per renderer-fed step, pose checksum %.9f, all equal: %s" %
      (n_obj, dt * 1e3, float(np.abs(poses).sum()), bool(all(np.array_equal(p, poses[0]) for p in poses))))
