"""The focused renderers' device path restated on the host with the device's own arithmetic (csrc/m3t_raster.h: set-up,
survivor list, row scan in the pieces the kernels cut it into, minimum of the packed words) against the ORACLE's
focused renderer on the scene of tools/raster_probe.py (prism + bottle of 20 950 triangles, both cameras): depth and
silhouette images equal, pixel for pixel."""
import ctypes as C
import os
import subprocess

import numpy as np

import golden_scene as gs
import util
from util import host


def test_host_restatement_of_the_device_rasteriser_equals_the_oracle(tmp_path):
    so = str(tmp_path / "libraster_stats.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so,
                    os.path.join(util.ROOT, "tests", "cpp", "raster_stats.cpp")], check=True)
    lib = C.CDLL(so)
    api = util.open_oracle()
    f = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                          depth_params=dict(n_unoccluded_iterations=0))
    geometry, schauma = gs.fixture_renderer_geometry(api, f.body)
    S = 200
    load_obj = util.pkg.config.load_obj
    bodies = []
    for order, (name, body2world, g2b, body_id) in enumerate((
            ("triangle", f.body.body2world_pose(), np.asarray(gs.mtv.GEOMETRY2BODY, np.float32), 150),
            ("schauma", np.linalg.inv(gs.SCHAUMA_WORLD2BODY.astype(np.float64)).astype(np.float32),
             np.asarray(gs.SCHAUMA_GEOMETRY2BODY, np.float32), 50))):
        v, t = load_obj(os.path.join(util.GOLDEN, "_body/%s.obj" % name))
        bodies.append((order, np.ascontiguousarray(v, np.float32), np.ascontiguousarray(t, np.int32),
                       np.asarray(body2world, np.float32), g2b.reshape(4, 4), body_id))
    f32 = np.float32
    for camera, k, w2c in ((f.color_camera, gs.mtv.COLOR_INTRINSICS, np.eye(4, dtype=f32)),
                           (f.depth_camera, gs.mtv.DEPTH_INTRINSICS, np.linalg.inv(gs.mtv.DEPTH_CAMERA2WORLD).astype(f32))):
        r = host.FocusedSilhouetteRenderer(api, geometry, camera, id_type=0, image_size=S)
        r.AddReferencedBody(f.body)
        r.StartRendering()
        depth, sil, corner_u, corner_v, scale, n_visible = r.images()
        assert n_visible == 1
        # FocusedRenderer::CalculateProjectionMatrix renderer.cpp:348-405 in float32, operation for operation like
        # focused_projection (the crop's side d itself is needed: S / scale does not give it back to the last bit)
        z_min, z_max = f32(0.02), f32(10.0)
        # Body::CalculateMaximumBodyDiameter body.cpp:244-250: over the vertices in the body frame
        tv, _ = load_obj(os.path.join(util.GOLDEN, "_body/triangle.obj"))
        m = np.asarray(gs.mtv.GEOMETRY2BODY, f32).reshape(4, 4)
        max_radius = f32(0.0)
        for vert in np.asarray(tv, f32):
            q = [m[kk, 3] + ((m[kk, 0] * vert[0] + m[kk, 1] * vert[1]) + m[kk, 2] * vert[2]) for kk in range(3)]
            max_radius = max(max_radius, np.sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2]), dtype=f32))
        rr = f32(0.5) * (f32(2.0) * max_radius)
        b2w = np.asarray(f.body.body2world_pose(), f32)
        t = b2w[:3, 3]
        x = ((w2c[0, 0] * t[0] + w2c[0, 1] * t[1]) + w2c[0, 2] * t[2]) + w2c[0, 3]
        y = ((w2c[1, 0] * t[0] + w2c[1, 1] * t[1]) + w2c[1, 2] * t[2]) + w2c[1, 3]
        z = ((w2c[2, 0] * t[0] + w2c[2, 1] * t[1]) + w2c[2, 2] * t[2]) + w2c[2, 3]
        x2, y2, z2, r2 = x * x, y * y, z * z, rr * rr
        rz = rr * z
        z2_r2 = z2 - r2
        z3_zr2 = z2_r2 * z
        fu, fv = f32(k["fu"]), f32(k["fv"])
        r_u = fu * (np.abs(x) * r2 + rz * np.sqrt(z2_r2 + x2, dtype=f32)) / z3_zr2
        r_v = fv * (np.abs(y) * r2 + rz * np.sqrt(z2_r2 + y2, dtype=f32)) / z3_zr2
        center_u = x * fu / z + f32(k["ppu"])
        center_v = y * fv / z + f32(k["ppv"])
        u_min, u_max, v_min, v_max = center_u - r_u, center_u + r_u, center_v - r_v, center_v + r_v
        d = np.maximum(u_max - u_min, v_max - v_min) * f32(1.05)
        assert f32(0.5) * (u_min + u_max - d) == f32(corner_u) and f32(S) / d == f32(scale)  # the oracle's crop
        P = np.zeros((4, 4), f32)
        ppu_scaled = (f32(k["ppu"]) - f32(corner_u)) * f32(scale)
        ppv_scaled = (f32(k["ppv"]) - f32(corner_v)) * f32(scale)
        P[0, 0] = f32(2.0) * f32(k["fu"]) / d
        P[0, 2] = f32(2.0) * (ppu_scaled + f32(0.5)) / f32(S) - f32(1.0)
        P[1, 1] = f32(2.0) * f32(k["fv"]) / d
        P[1, 2] = f32(2.0) * (ppv_scaled + f32(0.5)) / f32(S) - f32(1.0)
        P[2, 2] = (z_max + z_min) / (z_max - z_min)
        P[2, 3] = f32(-2.0) * z_max * z_min / (z_max - z_min)
        P[3, 2] = f32(1.0)
        packed = np.full(S * S, 0xFFFFFFFF, np.uint32)

        def mul44(a, b):  # the kernels' mul44: ((a0 b0 + a1 b1) + a2 b2) + a3 b3 per element, float32
            out = np.zeros((4, 4), f32)
            for c in range(4):
                for rr in range(4):
                    out[rr, c] = ((a[rr, 0] * b[0, c] + a[rr, 1] * b[1, c]) + a[rr, 2] * b[2, c]) + a[rr, 3] * b[3, c]
            return out

        for order, v, t, b2w, g2b, body_id in bodies:
            trans = mul44(P, mul44(w2c, mul44(b2w, g2b.astype(f32))))
            low = (order << 8) | body_id
            lib.raster_body(np.ascontiguousarray(trans.T).ctypes.data_as(C.POINTER(C.c_float)),
                            v.ctypes.data_as(C.POINTER(C.c_float)), t.ctypes.data_as(C.POINTER(C.c_int)), len(t), 1, S,
                            C.c_uint(low), packed.ctypes.data_as(C.POINTER(C.c_uint)))
        got_depth = np.where(packed == 0xFFFFFFFF, 65535, packed >> 16).astype(np.uint16).reshape(S, S)
        got_sil = np.where(packed == 0xFFFFFFFF, 0, packed & 0xFF).astype(np.uint8).reshape(S, S)
        assert np.array_equal(got_depth, depth)
        assert np.array_equal(got_sil, sil)
        assert (depth != 65535).sum() > 10000
