"""The reference's modality-test fixture (test/common_test.cpp:6-15,96-140, test/modality_test.cpp)
behind one C-ABI context: triangle body at the test pose, frame 200 of data/_sequence, default
modality parameters, and the regenerated template views of tests/golden/triangle_views.npz."""
import os
import sys

import numpy as np

import util
from util import host

sys.path.insert(0, os.path.join(util.ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(util.ROOT, "oracle"))
import make_triangle_views as mtv  # noqa: E402


def load_png(rel):
    """BGR8 as cv::imread gives it, or the 16-bit depth image (cv::IMREAD_UNCHANGED)"""
    from PIL import Image
    a = np.array(Image.open(os.path.join(util.GOLDEN, rel)))
    return np.ascontiguousarray(a[..., ::-1]) if a.ndim == 3 else a


def views():
    return np.load(os.path.join(util.ROOT, "tests", "golden", "triangle_views.npz"))


class RegionFixture:
    def __init__(self, api, measure_occlusions=False, **params):
        v = views()
        self.api = api
        self.image = load_png("_sequence/color_camera_image_200.png")
        self.model = host.RegionModel(api, data_points=v["region_points"], orientations=v["region_orientations"],
                                      contour_lengths=v["region_contour_lengths"])
        self.body = host.Body(api, mtv.body2world())
        self.camera = host.ColorCamera(api, **mtv.COLOR_INTRINSICS)
        if measure_occlusions:  # RegionModalityTest.CalculateCorrespondencesMeasuredOcclusions
            w2c = np.linalg.inv(mtv.DEPTH_CAMERA2WORLD).astype(np.float32)
            self.depth_camera = host.DepthCamera(api, depth_scale=mtv.DEPTH_SCALE, world2camera_pose=w2c,
                                                 **mtv.DEPTH_INTRINSICS)
            self.depth_camera.UpdateImage(load_png("_sequence/depth_camera_image_200.png"))
            self.modality = host.RegionModality(api, self.body, self.camera, self.model,
                                                depth_camera=self.depth_camera, measure_occlusions=1,
                                                n_unoccluded_iterations=0)
        else:
            self.modality = host.RegionModality(api, self.body, self.camera, self.model, **params)  # else defaults
        self.optimizer = host.Optimizer(api, body=self.body, modalities=[self.modality])
        self.camera.UpdateImage(self.image)
        self.tracker = host.Tracker(api)


class DepthFixture:
    def __init__(self, api, measure_occlusions=False, **params):
        v = views()
        self.api = api
        self.image = load_png("_sequence/depth_camera_image_200.png")
        self.model = host.DepthModel(api, data_points=v["depth_points"], orientations=v["depth_orientations"],
                                     surface_areas=v["depth_surface_areas"])
        self.body = host.Body(api, mtv.body2world())
        w2c = np.linalg.inv(mtv.DEPTH_CAMERA2WORLD).astype(np.float32)
        self.camera = host.DepthCamera(api, depth_scale=mtv.DEPTH_SCALE, world2camera_pose=w2c,
                                       **mtv.DEPTH_INTRINSICS)
        kw = dict(measure_occlusions=1, n_unoccluded_iterations=0) if measure_occlusions else dict(params)
        self.body2camera = (w2c @ mtv.body2world()).astype(np.float32)
        self.modality = host.DepthModality(api, self.body, self.camera, self.model, **kw)  # else defaults
        self.optimizer = host.Optimizer(api, body=self.body, modalities=[self.modality])
        self.camera.UpdateImage(self.image)
        self.tracker = host.Tracker(api)


def golden(name):
    return util.read_golden_matrix("modality_test/" + name)


def scaled_error(h, h_golden):
    """|dH_ij| relative to sqrt(|H_ii H_jj|): the reference's element-wise relative test
    (common_test.cpp:206-226) blows up on elements that are cancellation residues (H_01 = 0.23
    beside diagonals of 70) -- this is the same test with the natural scale of the element"""
    d = np.sqrt(np.abs(np.diag(h_golden)))
    return float(np.max(np.abs(h - h_golden) / np.outer(d, d)))


def render_lines_visualisation(image, histogram_f, histogram_b, lines, n_bins, scale, distribution_length):
    """RegionModality::VisualizeLines region_modality.cpp:1720-1803: probability image + the
    correspondence lines coloured by their distribution (what region_modality.png holds)"""
    f32 = np.float32
    shift = 8 - int(np.log2(n_bins))
    idx = ((image[..., 0].astype(np.int64) >> shift) * n_bins * n_bins +
           (image[..., 1].astype(np.int64) >> shift) * n_bins + (image[..., 2].astype(np.int64) >> shift))
    pf, pb = histogram_f[idx], histogram_b[idx]
    s = (pf + pb).astype(f32)
    pbn = np.where((pf != 0) | (pb != 0), pb / np.where(s == 0, 1, s), f32(0.5)).astype(f32)
    vis = np.repeat(np.rint(f32(255.0) * pbn).astype(np.int32)[..., None], 3, axis=2)
    color_line = np.array([24, 184, 234], f32)
    color_hp = np.array([61, 63, 179], f32)
    fscale = f32(scale)
    for l in lines:
        nu, nv = f32(l["normal_u"]), f32(l["normal_v"])
        if abs(nu) > abs(nv):
            us, vs = f32(np.sign(nu)), nv / abs(nu)
        else:
            us, vs = nu / abs(nv), f32(np.sign(nv))
        x = -fscale * f32((distribution_length - 1) / 2) - (fscale - f32(1)) / f32(2)
        u = f32(l["center_u"]) + us * x + f32(0.5)
        v = f32(l["center_v"]) + vs * x + f32(0.5)
        for i in range(distribution_length):
            r = min(f32(3) * f32(l["distribution"][i]), f32(1))
            c = np.rint(r * color_hp) + np.rint((f32(1) - r) * color_line)
            for _ in range(scale):
                vis[int(v), int(u)] = c
                u = f32(u + us)
                v = f32(v + vs)
    return vis


class TrackerFixture:
    """TrackerTest / RefinerTest / OptimizerTest set-up (test/tracker_test.cpp:25-41): both
    modalities of the triangle on one link"""

    def __init__(self, api, measure_occlusions, tikhonov_rotation=1000.0, tikhonov_translation=30000.0,
                 n_corr_iterations=7, n_update_iterations=2, region_params=None, depth_params=None):
        v = views()
        self.region_model = host.RegionModel(api, data_points=v["region_points"],
                                             orientations=v["region_orientations"],
                                             contour_lengths=v["region_contour_lengths"])
        self.depth_model = host.DepthModel(api, data_points=v["depth_points"], orientations=v["depth_orientations"],
                                           surface_areas=v["depth_surface_areas"])
        self.body = host.Body(api, mtv.body2world())
        self.color_camera = host.ColorCamera(api, **mtv.COLOR_INTRINSICS)
        w2c = np.linalg.inv(mtv.DEPTH_CAMERA2WORLD).astype(np.float32)
        self.depth_camera = host.DepthCamera(api, depth_scale=mtv.DEPTH_SCALE, world2camera_pose=w2c,
                                             **mtv.DEPTH_INTRINSICS)
        if measure_occlusions:
            self.region = host.RegionModality(api, self.body, self.color_camera, self.region_model,
                                              depth_camera=self.depth_camera, measure_occlusions=1)
            self.depth = host.DepthModality(api, self.body, self.depth_camera, self.depth_model, measure_occlusions=1)
        else:
            self.region = host.RegionModality(api, self.body, self.color_camera, self.region_model,
                                              **(region_params or {}))
            self.depth = host.DepthModality(api, self.body, self.depth_camera, self.depth_model,
                                            **(depth_params or {}))
        self.optimizer = host.Optimizer(api, body=self.body, modalities=[self.region, self.depth],
                                        tikhonov_parameter_rotation=tikhonov_rotation,
                                        tikhonov_parameter_translation=tikhonov_translation)
        self.color_camera.UpdateImage(load_png("_sequence/color_camera_image_200.png"))
        self.depth_camera.UpdateImage(load_png("_sequence/depth_camera_image_200.png"))
        self.tracker = host.Tracker(api, n_corr_iterations, n_update_iterations)


def point_mask(shape, points_f_body, body2camera, intrinsics):
    """DrawPointInImage common.cpp:268-276 for every point: cv::circle(radius 1, FILLED) = the pixel and
    its four neighbours"""
    f32 = np.float32
    mask = np.zeros(shape, bool)
    for p in points_f_body:
        c = (body2camera[:3, :3] @ p + body2camera[:3, 3]).astype(f32)
        u = int(c[0] * f32(intrinsics["fu"]) / c[2] + f32(intrinsics["ppu"]) + 0.5)
        v = int(c[1] * f32(intrinsics["fv"]) / c[2] + f32(intrinsics["ppv"]) + 0.5)
        for du, dv in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
            if 0 <= v + dv < shape[0] and 0 <= u + du < shape[1]:
                mask[v + dv, u + du] = True
    return mask


def golden_point_mask(rel, bgr):
    return (load_png(rel).astype(np.int32) == np.asarray(bgr)).all(axis=2)


# ---- renderer-fed branches: the two fixture bodies with their meshes (test/common_test.cpp:6-39,41-94) ----
SCHAUMA_WORLD2BODY = np.array([[0.607676, 0.408914, -0.680823, 0.297794], [0.786584, -0.428213, 0.444880, -0.189009],
                               [-0.109620, -0.805867, -0.581860, 0.255284], [0, 0, 0, 1]], np.float32)
SCHAUMA_GEOMETRY2BODY = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -0.097], [0, 0, 0, 1]], np.float32)


def fixture_renderer_geometry(api, triangle_body):
    """RendererGeometry holding the triangle (body id 150, region id 150) and the schauma bottle
    (body id 50, region id 150), data/_body/{triangle,schauma}.yaml"""
    load_obj = util.pkg.config.load_obj  # (bench.py's extras leg comes through here: no checker code on that path)
    tv, tf = load_obj(os.path.join(util.GOLDEN, "_body/triangle.obj"))
    sv, sf = load_obj(os.path.join(util.GOLDEN, "_body/schauma.obj"))
    triangle_body.set_geometry(tv, tf, np.asarray(mtv.GEOMETRY2BODY, np.float32), body_id=150, region_id=150)
    schauma = host.Body(api, np.linalg.inv(SCHAUMA_WORLD2BODY.astype(np.float64)).astype(np.float32))
    schauma.set_geometry(sv, sf, SCHAUMA_GEOMETRY2BODY, body_id=50, region_id=150)
    geometry = host.RendererGeometry(api)
    geometry.AddBody(triangle_body)
    geometry.AddBody(schauma)
    return geometry, schauma


def focused_point_mask(size, points_f_body, body2camera, intrinsics, corner_u, corner_v, scale):
    """DrawFocusedPointInImage common.cpp:293-304"""
    f32 = np.float32
    mask = np.zeros((size, size), bool)
    for p in points_f_body:
        c = (body2camera[:3, :3] @ p + body2camera[:3, 3]).astype(f32)
        u = c[0] * f32(intrinsics["fu"]) / c[2] + f32(intrinsics["ppu"])
        v = c[1] * f32(intrinsics["fv"]) / c[2] + f32(intrinsics["ppv"])
        uf = int((u - f32(corner_u)) * f32(scale) + f32(0.5))
        vf = int((v - f32(corner_v)) * f32(scale) + f32(0.5))
        for du, dv in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
            if 0 <= vf + dv < size and 0 <= uf + du < size:
                mask[vf + dv, uf + du] = True
    return mask
