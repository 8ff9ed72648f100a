"""Timings of the renderer-fed configuration (python tools/render_bench.py, on the GPU box):
one focused rendering of the triangle + schauma bottle (20 958 triangles, 200 x 200) and a whole
tracking step of Region + Depth modality with region checking, silhouette checking and modelled
occlusions (7 x 2 iterations, renderings refreshed before every correspondence search)."""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401

torch.cuda.init()
import golden_scene as gs  # noqa: E402
import util  # noqa: E402
from util import host  # noqa: E402

api = util.open_hip()
f = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                      depth_params=dict(n_unoccluded_iterations=0))
geometry, schauma = gs.fixture_renderer_geometry(api, f.body)
cd = host.FocusedBasicDepthRenderer(api, geometry, f.color_camera)
cs = host.FocusedSilhouetteRenderer(api, geometry, f.color_camera, id_type=1)
dd = host.FocusedBasicDepthRenderer(api, geometry, f.depth_camera)
ds = host.FocusedSilhouetteRenderer(api, geometry, f.depth_camera, id_type=0)
for r in (cd, cs, dd, ds):
    r.AddReferencedBody(f.body)
cd.StartRendering()
t0 = time.perf_counter()
for _ in range(50):
    cd.StartRendering()  # includes a stream synchronisation
print("one focused rendering incl. launch + sync: %.1f us" % ((time.perf_counter() - t0) / 50 * 1e6))
f.region.ModelOcclusions(cd)
f.region.UseRegionChecking(cs)
f.depth.ModelOcclusions(dd)
f.depth.UseSilhouetteChecking(ds)
start = f.body.body2world_pose()
f.tracker.StartModalities(0)
f.tracker.ExecuteTrackingStep(0)
api.call("sync")
t0 = time.perf_counter()
n = 20
for _ in range(n):
    f.body.set_body2world_pose(start)
    f.tracker.ExecuteTrackingStep(0)
api.call("sync")
print("tracking step with all renderer-fed branches (1 object, 4 renderers): %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
