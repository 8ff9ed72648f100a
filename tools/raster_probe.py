"""Developer tool: wall time of one focused rendering (clear + rasterise + unpack + stream synchronisation; triangle +
schauma bottle, 20 958 triangles, 200 x 200, the scene of tools/render_bench.py) for several builds of the library,
and a checksum of the depth / silhouette images (builds that only re-balance the rasteriser must agree).

With --step also the tracking step of tools/render_bench.py (Region + Depth, region / silhouette checking and
modelled occlusions: four focused renderings before every correspondence search), without torch.

  python tools/raster_probe.py [--step] lib_a.so [lib_b.so ...]"""
import os
import sys
import time
import zlib

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_scene as gs  # noqa: E402
import util  # noqa: E402
from util import host  # noqa: E402


def main():
    step = "--step" in sys.argv
    for lib in [a for a in sys.argv[1:] if a != "--step"]:
        api = util.pkg.CApi(lib, "m3t_hip_")
        f = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                              depth_params=dict(n_unoccluded_iterations=0))
        geometry, schauma = gs.fixture_renderer_geometry(api, f.body)
        cd = host.FocusedBasicDepthRenderer(api, geometry, f.color_camera)
        cs = host.FocusedSilhouetteRenderer(api, geometry, f.color_camera, id_type=1)
        for r in (cd, cs):
            r.AddReferencedBody(f.body)
        out = []
        for r in (cd, cs):
            r.StartRendering()
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(40):
                    r.StartRendering()  # includes a stream synchronisation
                best = min(best, (time.perf_counter() - t0) / 40 * 1e6)
            depth, sil = r.images()[:2]
            crc = zlib.crc32(depth.tobytes())
            if sil is not None:
                crc = zlib.crc32(sil.tobytes(), crc)
            out.append("%.1f us (crc %08x)" % (best, crc))
        print("%-40s depth renderer %s   silhouette renderer %s" % (os.path.relpath(lib, ROOT), out[0], out[1]), flush=True)
        if step:
            dd = host.FocusedBasicDepthRenderer(api, geometry, f.depth_camera)
            ds = host.FocusedSilhouetteRenderer(api, geometry, f.depth_camera, id_type=0)
            for r in (dd, ds):
                r.AddReferencedBody(f.body)
            f.region.ModelOcclusions(cd)
            f.region.UseRegionChecking(cs)
            f.depth.ModelOcclusions(dd)
            f.depth.UseSilhouetteChecking(ds)
            start = f.body.body2world_pose()
            f.tracker.StartModalities(0)
            f.tracker.ExecuteTrackingStep(0)
            api.call("sync")
            end = f.body.body2world_pose()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(20):
                    f.body.set_body2world_pose(start)
                    f.tracker.ExecuteTrackingStep(0)
                api.call("sync")
                best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
            print("%-40s tracking step with all renderer-fed branches (1 object, 4 renderers): %.3f ms   pose crc %08x" %
                  ("", best, zlib.crc32(end.tobytes())), flush=True)


if __name__ == "__main__":
    main()
