"""Regenerates the two template views of data/_body/triangle.obj that the reference's modality
tests use (RegionModalityTest / DepthModalityTest, test/modality_test.cpp:17-34,318-337): the
closest of the 2562 views of the default models (sphere_radius 0.8, n_divides 4, 200 points,
image_size 2000) for the test pose and its four nearest neighbours, through the OpenGL-free generator gl_model.py.
Output: tests/golden/triangle_views.npz (committed; tests/test_modality_goldens.py reads it and
tests/test_model_generation.py checks that this script still reproduces it).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

REF = os.path.join(HERE, "reference")
# test/common_test.cpp:6-15 (world2body), data/_body/triangle.yaml (geometry2body)
WORLD2BODY = np.array([[0.607676, 0.408914, -0.680823, 0.472944], [0.786584, -0.428213, 0.444880, -0.213009],
                       [-0.109620, -0.805867, -0.581860, 0.346384], [0, 0, 0, 1]], np.float32)
GEOMETRY2BODY = [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -0.006], [0, 0, 0, 1]]
# data/_sequence/{color,depth}_camera.yaml
COLOR_INTRINSICS = dict(fu=698.128, fv=698.617, ppu=478.459, ppv=274.426, width=960, height=540)
DEPTH_INTRINSICS = dict(fu=425.773, fv=425.773, ppu=427.202, ppv=237.662, width=848, height=480)
DEPTH_SCALE = 0.001
DEPTH_CAMERA2WORLD = np.array([[0.99985489, 0.00778240, 0.01509715, 0.01453388],
                               [-0.00782678, 0.99996543, 0.00288261, 0.00013995],
                               [-0.01507424, -0.00300036, 0.99988175, 0.00051057], [0, 0, 0, 1]], np.float64)
SPHERE_RADIUS, N_DIVIDES, N_POINTS, IMAGE_SIZE = 0.8, 4, 200, 2000


def body2world():
    return np.linalg.inv(WORLD2BODY.astype(np.float64)).astype(np.float32)


def closest_views(poses, body2camera, n):
    """RegionModel::GetClosestView region_model.cpp:105-130: the n best views, best first"""
    b2c = np.asarray(body2camera, np.float64)
    o = np.linalg.inv(b2c[:3, :3]) @ b2c[:3, 3]
    o /= np.linalg.norm(o)
    dots = np.asarray([p[:3, 2] for p in poses]) @ o.astype(np.float32)
    return [int(v) for v in np.argsort(-dots)[:n]]


N_VIEWS = 5  # the closest view and the neighbours a tracking step can switch to


def generate():
    # the checker is imported here, not at module level: the constants above are also read by bench.py's extras
    # leg (through tests/golden_scene.py), which must not load anything under oracle/
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle"))
    import gl_model as g
    body = g.ConvexBody(os.path.join(REF, "_body/triangle.obj"), GEOMETRY2BODY)
    poses = g.geodesic_poses(N_DIVIDES, SPHERE_RADIUS)
    b2w = body2world().astype(np.float64)
    rv = closest_views(poses, b2w, N_VIEWS)  # color camera2world is the identity
    dv = closest_views(poses, np.linalg.inv(DEPTH_CAMERA2WORLD) @ b2w, N_VIEWS)
    R = [g.region_view(body, poses[v], SPHERE_RADIUS, N_POINTS, IMAGE_SIZE) for v in rv]
    D = [g.depth_view(body, poses[v], SPHERE_RADIUS, N_POINTS, IMAGE_SIZE) for v in dv]
    return dict(region_views=np.asarray(rv), region_points=np.stack([r[0] for r in R]),
                region_orientations=np.stack([r[1] for r in R]),
                region_contour_lengths=np.asarray([r[2] for r in R], np.float32),
                depth_views=np.asarray(dv), depth_points=np.stack([d[0] for d in D]),
                depth_orientations=np.stack([d[1] for d in D]),
                depth_surface_areas=np.asarray([d[2] for d in D], np.float32), n_views=len(poses))


if __name__ == "__main__":
    out = generate()
    np.savez_compressed(os.path.join(HERE, "triangle_views.npz"), **out)
    print("views", out["region_views"], out["depth_views"], "of", out["n_views"], "contour lengths",
          out["region_contour_lengths"], "surface areas", out["depth_surface_areas"])
