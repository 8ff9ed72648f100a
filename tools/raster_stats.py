"""Developer tool (CPU only): what the focused rasteriser has to distribute on the scene of tools/raster_probe.py --
per body, how many triangles survive set-up, how large their bounding boxes are and how many pixels they cover --
from the oracle's crop of the same rendering and csrc/m3t_raster.h compiled for the host.

  python tools/raster_stats.py"""
import ctypes as C
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_scene as gs  # noqa: E402
import util  # noqa: E402
from util import host  # noqa: E402


def main():
    so = os.path.join(tempfile.mkdtemp(), "libraster_stats.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                    os.path.join(ROOT, "tests", "cpp", "raster_stats.cpp")], check=True)
    lib = C.CDLL(so)
    api = util.open_oracle()
    f = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                          depth_params=dict(n_unoccluded_iterations=0))
    geometry, schauma = gs.fixture_renderer_geometry(api, f.body)
    S = 200
    for name, camera in (("colour camera", f.color_camera), ("depth camera", f.depth_camera)):
        r = host.FocusedBasicDepthRenderer(api, geometry, camera, image_size=S)
        r.AddReferencedBody(f.body)
        r.StartRendering()
        depth, _, corner_u, corner_v, scale, n_visible = r.images()
        intr = camera.intrinsics if hasattr(camera, "intrinsics") else None
        k = gs.mtv.COLOR_INTRINSICS if name.startswith("colour") else gs.mtv.DEPTH_INTRINSICS
        z_min, z_max = 0.02, 10.0
        d = S / scale
        P = np.zeros((4, 4), np.float32)
        ppu_scaled = (np.float32(k["ppu"]) - np.float32(corner_u)) * np.float32(scale)
        ppv_scaled = (np.float32(k["ppv"]) - np.float32(corner_v)) * np.float32(scale)
        P[0, 0] = 2.0 * k["fu"] / d
        P[0, 2] = 2.0 * (ppu_scaled + 0.5) / S - 1.0
        P[1, 1] = 2.0 * k["fv"] / d
        P[1, 2] = 2.0 * (ppv_scaled + 0.5) / S - 1.0
        P[2, 2] = (z_max + z_min) / (z_max - z_min)
        P[2, 3] = -2.0 * z_max * z_min / (z_max - z_min)
        P[3, 2] = 1.0
        w2c = np.linalg.inv(gs.mtv.DEPTH_CAMERA2WORLD).astype(np.float32) if name.startswith("depth") else np.eye(4, dtype=np.float32)
        print("%s: crop corner (%.1f, %.1f) scale %.3f, %d of %d pixels rendered" %
              (name, corner_u, corner_v, scale, int((depth != 65535).sum()), S * S))
        load_obj = util.pkg.config.load_obj
        for body_name, body2world, g2b in (("triangle", f.body.body2world_pose(), np.asarray(gs.mtv.GEOMETRY2BODY, np.float32)),
                                           ("schauma", np.linalg.inv(gs.SCHAUMA_WORLD2BODY.astype(np.float64)).astype(np.float32),
                                            np.asarray(gs.SCHAUMA_GEOMETRY2BODY, np.float32))):
            v, t = load_obj(os.path.join(util.GOLDEN, "_body/%s.obj" % body_name))
            trans = (P @ w2c @ np.asarray(body2world, np.float32) @ g2b.reshape(4, 4)).astype(np.float32)
            out = (C.c_longlong * 40)()
            vv = np.ascontiguousarray(v, np.float32)
            tt = np.ascontiguousarray(t, np.int32)
            lib.raster_stats(np.ascontiguousarray(trans.T).ctypes.data_as(C.POINTER(C.c_float)),
                             vv.ctypes.data_as(C.POINTER(C.c_float)), tt.ctypes.data_as(C.POINTER(C.c_int)), len(tt), 1, S, out)
            o = list(out)
            print("  %-9s %6d triangles, %6d survive set-up; box pixels %8d, covered %7d; boxes > 192 px: %5d, largest %6d" %
                  (body_name, len(tt), o[0], o[1], o[2], o[3], o[4]))
            print("            boxes by size 1,2,4,..: %s" % o[5:21])
            print("            covered per triangle 1,2,4,..: %s" % o[21:37])


if __name__ == "__main__":
    main()
