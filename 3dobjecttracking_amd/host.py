"""Host-side mirror of the reference's object graph for the hot path.

Same class and method names as M3T (Body, ColorCamera, DepthCamera,
RegionModel, DepthModel, RegionModality, DepthModality, Link, Optimizer,
Constraint, Tracker with StartModalities / CalculateCorrespondences /
CalculateGradientAndHessian / CalculateOptimization / CalculateResults /
ExecuteTrackingStep; M3T/include/m3t/tracker.h:131-160), but every object is a
handle into ONE batched device context behind the C-ABI: a Tracker call runs
the step for all registered modalities in one launch sequence.

Error behaviour follows the reference: steps return bool (False + message on
stderr when something is not set up), constructors raise on invalid arguments.
"""
import ctypes as C
import sys

import numpy as np

from . import _capi
from ._capi import (DATA_LINE_DTYPE, DATA_POINT_DTYPE, DepthModalityParams, DepthModelDesc, Intrinsics, M3TError,
                    RegionModalityParams, RegionModelDesc, fptr, iptr, pose_arg, pose_ret)


class Tracker:
    """m3t::Tracker restricted to the tracking step (tracker.cpp:344-364, 430-517)."""

    def __init__(self, api, n_corr_iterations=5, n_update_iterations=2):
        self.api = api
        self.n_corr_iterations = n_corr_iterations
        self.n_update_iterations = n_update_iterations
        api.call("tracker_set_iterations", n_corr_iterations, n_update_iterations)

    def _step(self, name, *args):
        rc = self.api.raw(name, *args)
        if rc < 0:
            sys.stderr.write(self.api.last_error() + "\n")
            return False
        return True

    def StartModalities(self, iteration):
        return self._step("start_modalities", iteration)

    def CalculateCorrespondences(self, iteration, corr_iteration):
        return self._step("calculate_correspondences", iteration, corr_iteration)

    def CalculateGradientAndHessian(self, iteration, corr_iteration, update_iteration):
        return self._step("calculate_gradient_and_hessian", iteration, corr_iteration, update_iteration)

    def CalculateOptimization(self, iteration, corr_iteration, update_iteration):
        return self._step("calculate_optimization", iteration, corr_iteration, update_iteration)

    def CalculateResults(self, iteration):
        return self._step("calculate_results", iteration)

    def CalculateConsistentPoses(self):
        return self._step("calculate_consistent_poses")

    def CalculateOptimizationBegin(self):
        """first half of CalculateOptimization: the link sums of this process's modalities, 6 + 36 floats per link.
        Returns (pointer, count); device pointer for the HIP library, host pointer for the oracle."""
        ptr = _capi.c_float_p()
        n = C.c_size_t()
        self.api.call("calculate_optimization_begin", C.byref(ptr), C.byref(n))
        return ptr, n.value

    def CalculateOptimizationEnd(self):
        return self._step("calculate_optimization_end")

    def ExecuteTrackingStep(self, iteration):
        return self._step("execute_tracking_step", iteration)

    # asynchronous ingest (HIP library only)
    def register_host_buffer(self, array):
        """page-lock a caller-owned frame buffer so that asynchronous uploads overlap the tracking kernels"""
        self.api.call("host_register", array.ctypes.data_as(C.c_void_p), array.nbytes)

    def unregister_host_buffer(self, array):
        self.api.call("host_unregister", array.ctypes.data_as(C.c_void_p))

    def select_slot(self, slot):
        self.api.call("cameras_select_slot", slot)

    def ingest_sync(self):
        self.api.call("ingest_sync")

    def RefinePoses(self, n_corr_iterations=7, n_update_iterations=2):
        """m3t::Refiner::RefinePoses (refiner.cpp:76-117)"""
        return self._step("refine_poses", n_corr_iterations, n_update_iterations)

    def ExecuteTrackingCycle(self, iteration):  # ICG/RBGT/SRT3D name
        return self._step("execute_tracking_cycle", iteration)

    def Sync(self):
        return self._step("sync")


class Body:
    def __init__(self, api, body2world_pose=np.eye(4)):
        self.api = api
        self.id = api.call("body_create", fptr(pose_arg(body2world_pose)))

    def set_body2world_pose(self, pose):
        self.api.call("body_set_body2world_pose", self.id, fptr(pose_arg(pose)))

    def body2world_pose(self):
        buf = np.zeros(16, np.float32)
        self.api.call("body_get_body2world_pose", self.id, fptr(buf))
        return pose_ret(buf)

    def set_geometry(self, vertices, triangles, geometry2body_pose=np.eye(4), geometry_counterclockwise=True,
                     geometry_enable_culling=True, body_id=0, region_id=0):
        """the mesh of body.h (vertices in metres) for the renderer-fed branches"""
        v = np.ascontiguousarray(vertices, np.float32)
        t = np.ascontiguousarray(triangles, np.int32)
        g = _capi.BodyGeometry()
        g.vertices, g.n_vertices = fptr(v), len(v)
        g.triangles, g.n_triangles = iptr(t), len(t)
        g2b = pose_arg(geometry2body_pose)
        for i in range(16):
            g.geometry2body[i] = float(g2b[i])
        g.geometry_counterclockwise = int(geometry_counterclockwise)
        g.geometry_enable_culling = int(geometry_enable_culling)
        g.body_id, g.region_id = int(body_id), int(region_id)
        self.api.call("body_set_geometry", self.id, C.byref(g))


class ColorHistograms:
    """m3t::ColorHistograms shared by several RegionModalities (color_histograms.h:36-40)"""

    def __init__(self, api, n_bins=16, learning_rate_f=0.2, learning_rate_b=0.2):
        self.api = api
        self.n_bins = n_bins
        self.id = api.call("color_histograms_create", n_bins, learning_rate_f, learning_rate_b)


class RendererGeometry:
    """m3t::RendererGeometry: the bodies a renderer draws, in draw order"""

    def __init__(self, api):
        self.api = api
        self.id = api.call("renderer_geometry_create")

    def AddBody(self, body):
        self.api.call("renderer_geometry_add_body", self.id, body.id)


class _FocusedRenderer:
    def __init__(self, api, rid, image_size, silhouette):
        self.api, self.id, self.image_size, self.silhouette = api, rid, image_size, silhouette

    def AddReferencedBody(self, body):
        self.api.call("renderer_add_referenced_body", self.id, body.id)

    def StartRendering(self):
        self.api.call("renderer_start_rendering", self.id)
        return True

    def images(self):
        """(depth u16 [S,S], silhouette u8 [S,S] or None, corner_u, corner_v, scale, n_visible)"""
        s = self.image_size
        depth = np.zeros((s, s), np.uint16)
        sil = np.zeros((s, s), np.uint8) if self.silhouette else None
        info = np.zeros(3, np.float32)
        n = C.c_int()
        self.api.call("renderer_get_images", self.id, depth.ctypes.data_as(C.POINTER(C.c_uint16)),
                      sil.ctypes.data_as(C.POINTER(C.c_uint8)) if sil is not None else None, fptr(info), C.byref(n))
        return depth, sil, float(info[0]), float(info[1]), float(info[2]), n.value


class FocusedBasicDepthRenderer(_FocusedRenderer):
    """basic_depth_renderer.h:120-128"""

    def __init__(self, api, renderer_geometry, camera, image_size=200, z_min=0.02, z_max=10.0):
        rid = api.call("focused_depth_renderer_create", renderer_geometry.id, camera.id, image_size, z_min, z_max)
        super().__init__(api, rid, image_size, False)


class FocusedSilhouetteRenderer(_FocusedRenderer):
    """silhouette_renderer.h:150-155; id_type 0 = IDType::BODY, 1 = IDType::REGION"""

    def __init__(self, api, renderer_geometry, camera, id_type=0, image_size=200, z_min=0.02, z_max=10.0):
        rid = api.call("focused_silhouette_renderer_create", renderer_geometry.id, camera.id, id_type, image_size,
                       z_min, z_max)
        super().__init__(api, rid, image_size, True)


class _Camera:
    def __init__(self, api, cam_id, width, height, bytes_per_pixel):
        self.api = api
        self.id = cam_id
        self.width, self.height, self.bpp = width, height, bytes_per_pixel

    def UpdateImage(self, image):
        """image: (H, W, 3) uint8 BGR or (H, W) uint16; row stride taken from the array."""
        a = np.asarray(image)
        assert a.shape[0] == self.height and a.shape[1] == self.width, (a.shape, self.height, self.width)
        assert a.itemsize * (a.shape[2] if a.ndim == 3 else 1) == self.bpp
        if a.strides[1] != self.bpp or (a.ndim == 3 and a.strides[2] != 1):
            a = np.ascontiguousarray(a)
        self.api.call("camera_upload", self.id, a.ctypes.data_as(C.c_void_p), a.strides[0])
        return True

    # device-side frame ring + asynchronous ingest (HIP library only)
    def set_ring(self, n_slots):
        self.api.call("camera_set_ring", self.id, n_slots)

    def upload_slot(self, slot, image, asynchronous=False):
        """Stage `image` into ring slot `slot`.  With asynchronous=True the call returns at once and
        the array must stay alive and unchanged until slot_sync(slot) / Tracker.ingest_sync() (page-lock it once with
        Tracker.register_host_buffer for a true overlapped copy)."""
        a = np.asarray(image)
        assert a.shape[0] == self.height and a.shape[1] == self.width
        assert a.strides[1] == self.bpp, "asynchronous uploads borrow the buffer: it must be pixel-contiguous"
        self.api.call("camera_upload_slot_async" if asynchronous else "camera_upload_slot", self.id, slot,
                      a.ctypes.data_as(C.c_void_p), a.strides[0])

    def select_slot(self, slot):
        self.api.call("camera_select_slot", self.id, slot)

    def slot_sync(self, slot):
        """wait until this camera's last asynchronous upload into `slot` has left its host buffer (and for no other
        copy: Tracker.ingest_sync waits for all cameras)"""
        self.api.call("camera_slot_sync", self.id, slot)

    def set_world2camera_pose(self, pose):
        self.api.call("camera_set_world2camera_pose", self.id, fptr(pose_arg(pose)))


def _intr(fu, fv, ppu, ppv, width, height):
    return Intrinsics(fu, fv, ppu, ppv, width, height)


class ColorCamera(_Camera):
    def __init__(self, api, fu, fv, ppu, ppv, width, height, world2camera_pose=np.eye(4)):
        i = _intr(fu, fv, ppu, ppv, width, height)
        cid = api.call("color_camera_create", C.byref(i), fptr(pose_arg(world2camera_pose)))
        super().__init__(api, cid, width, height, 3)


class DepthCamera(_Camera):
    def __init__(self, api, fu, fv, ppu, ppv, width, height, depth_scale, world2camera_pose=np.eye(4)):
        i = _intr(fu, fv, ppu, ppv, width, height)
        cid = api.call("depth_camera_create", C.byref(i), fptr(pose_arg(world2camera_pose)), depth_scale)
        super().__init__(api, cid, width, height, 2)
        self.depth_scale = depth_scale


class RegionModel:
    """Sparse viewpoint model for regions (runtime part of region_model.cpp)."""

    def __init__(self, api, path=None, data_points=None, orientations=None, contour_lengths=None,
                 stride_depth_offset=0.002, max_radius_depth_offset=0.05):
        self.api = api
        if path is not None:
            self.id = api.call("region_model_load", str(path).encode())
        else:
            dp = np.ascontiguousarray(data_points, np.float32)
            ori = np.ascontiguousarray(orientations, np.float32)
            cl = np.ascontiguousarray(contour_lengths, np.float32)
            assert dp.ndim == 3 and dp.shape[2] == _capi.M3T_REGION_POINT_FLOATS
            d = RegionModelDesc(dp.shape[0], dp.shape[1], fptr(dp), fptr(ori), fptr(cl), stride_depth_offset,
                                max_radius_depth_offset)
            self.id = api.call("region_model_create", C.byref(d))
        nv, npts, me = C.c_int(), C.c_int(), C.c_float()
        api.call("region_model_info", self.id, C.byref(nv), C.byref(npts), C.byref(me))
        self.n_views, self.n_points, self.max_contour_length = nv.value, npts.value, me.value

    def GetClosestView(self, body2camera_pose):
        v = C.c_int()
        self.api.call("region_model_closest_view", self.id, fptr(pose_arg(body2camera_pose)), C.byref(v))
        return v.value

    @classmethod
    def generate(cls, api, body, **params):
        """RegionModel::GenerateModel without OpenGL (HIP library only); body needs set_geometry()"""
        self = cls.__new__(cls)
        self.api = api
        # associated: (body, movable, same_region) triples of RegionModel::AddAssociatedBody
        associated = params.pop("associated", ())
        gp = C.byref(_capi.ModelGenerationParams(**params))
        if associated:
            ids = np.asarray([b.id for b, _, _ in associated], np.int32)
            mov = np.asarray([int(m) for _, m, _ in associated], np.int32)
            same = np.asarray([int(s) for _, _, s in associated], np.int32)
            self.id = api.call("region_model_generate_associated", body.id, gp, len(ids), iptr(ids), iptr(mov), iptr(same))
        else:
            self.id = api.call("region_model_generate", body.id, gp)
        nv, npts, me = C.c_int(), C.c_int(), C.c_float()
        api.call("region_model_info", self.id, C.byref(nv), C.byref(npts), C.byref(me))
        self.n_views, self.n_points, self.max_contour_length = nv.value, npts.value, me.value
        return self

    def views(self):
        pts = np.zeros((self.n_views, self.n_points, _capi.M3T_REGION_POINT_FLOATS), np.float32)
        ori = np.zeros((self.n_views, 3), np.float32)
        ext = np.zeros(self.n_views, np.float32)
        self.api.call("region_model_get_views", self.id, fptr(pts), fptr(ori), fptr(ext))
        return pts, ori, ext


class DepthModel:
    def __init__(self, api, path=None, data_points=None, orientations=None, surface_areas=None,
                 stride_depth_offset=0.002, max_radius_depth_offset=0.05):
        self.api = api
        if path is not None:
            self.id = api.call("depth_model_load", str(path).encode())
        else:
            dp = np.ascontiguousarray(data_points, np.float32)
            ori = np.ascontiguousarray(orientations, np.float32)
            sa = np.ascontiguousarray(surface_areas, np.float32)
            assert dp.ndim == 3 and dp.shape[2] == _capi.M3T_DEPTH_POINT_FLOATS
            d = DepthModelDesc(dp.shape[0], dp.shape[1], fptr(dp), fptr(ori), fptr(sa), stride_depth_offset,
                               max_radius_depth_offset)
            self.id = api.call("depth_model_create", C.byref(d))
        nv, npts, me = C.c_int(), C.c_int(), C.c_float()
        api.call("depth_model_info", self.id, C.byref(nv), C.byref(npts), C.byref(me))
        self.n_views, self.n_points, self.max_surface_area = nv.value, npts.value, me.value

    def GetClosestView(self, body2camera_pose):
        v = C.c_int()
        self.api.call("depth_model_closest_view", self.id, fptr(pose_arg(body2camera_pose)), C.byref(v))
        return v.value

    @classmethod
    def generate(cls, api, body, **params):
        """DepthModel::GenerateModel without OpenGL (HIP library only); body needs set_geometry()"""
        self = cls.__new__(cls)
        self.api = api
        occlusion_bodies = params.pop("occlusion_bodies", ())  # DepthModel::AddOcclusionBody
        gp = C.byref(_capi.ModelGenerationParams(**params))
        if occlusion_bodies:
            ids = np.asarray([b.id for b in occlusion_bodies], np.int32)
            self.id = api.call("depth_model_generate_occluded", body.id, gp, len(ids), iptr(ids))
        else:
            self.id = api.call("depth_model_generate", body.id, gp)
        nv, npts, me = C.c_int(), C.c_int(), C.c_float()
        api.call("depth_model_info", self.id, C.byref(nv), C.byref(npts), C.byref(me))
        self.n_views, self.n_points, self.max_surface_area = nv.value, npts.value, me.value
        return self

    def views(self):
        pts = np.zeros((self.n_views, self.n_points, _capi.M3T_DEPTH_POINT_FLOATS), np.float32)
        ori = np.zeros((self.n_views, 3), np.float32)
        ext = np.zeros(self.n_views, np.float32)
        self.api.call("depth_model_get_views", self.id, fptr(pts), fptr(ori), fptr(ext))
        return pts, ori, ext


class _Modality:
    def gradient_hessian(self):
        g = np.zeros(6, np.float32)
        h = np.zeros(36, np.float32)
        self.api.call("modality_get_gradient_hessian", self.id, fptr(g), fptr(h))
        return g, h.reshape(6, 6).T.copy()

    def gradient(self):
        return self.gradient_hessian()[0]

    def hessian(self):
        return self.gradient_hessian()[1]

    def set_gradient_hessian(self, g, h):
        g = np.ascontiguousarray(g, np.float32).reshape(6)
        h = np.ascontiguousarray(np.asarray(h, np.float32).reshape(6, 6).T).reshape(36)
        self.api.call("modality_set_gradient_hessian", self.id, fptr(g), fptr(h))


class RegionModality(_Modality):
    def __init__(self, api, body, color_camera, region_model, depth_camera=None, params=None, **kw):
        self.api = api
        self.params = params if params is not None else RegionModalityParams(**kw)
        self.id = api.call("region_modality_create", C.byref(self.params), body.id, color_camera.id,
                           region_model.id, depth_camera.id if depth_camera is not None else -1)
        self.n_bins = self.params.n_histogram_bins

    def UseSharedColorHistograms(self, color_histograms):
        self.api.call("region_modality_use_shared_color_histograms", self.id, color_histograms.id)
        self.n_bins = color_histograms.n_bins

    def ModelOcclusions(self, depth_renderer):
        self.api.call("region_modality_model_occlusions", self.id, depth_renderer.id)

    def UseRegionChecking(self, silhouette_renderer):
        self.api.call("region_modality_use_region_checking", self.id, silhouette_renderer.id)

    def data_lines(self):
        n = C.c_int()
        cap = self.params.n_lines_max
        out = np.zeros(cap, DATA_LINE_DTYPE)
        self.api.call("region_modality_get_lines", self.id, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        return out[:min(n.value, cap)]

    def histograms(self):
        n = self.n_bins ** 3
        f = np.zeros(n, np.float32)
        b = np.zeros(n, np.float32)
        self.api.call("region_modality_get_histograms", self.id, fptr(f), fptr(b))
        return f, b

    def set_histograms(self, f, b):
        f = np.ascontiguousarray(f, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        assert f.size == self.n_bins ** 3 and b.size == f.size
        self.api.call("region_modality_set_histograms", self.id, fptr(f), fptr(b))


class DepthModality(_Modality):
    def __init__(self, api, body, depth_camera, depth_model, params=None, **kw):
        self.api = api
        self.params = params if params is not None else DepthModalityParams(**kw)
        self.id = api.call("depth_modality_create", C.byref(self.params), body.id, depth_camera.id, depth_model.id)

    def ModelOcclusions(self, depth_renderer):
        self.api.call("depth_modality_model_occlusions", self.id, depth_renderer.id)

    def UseSilhouetteChecking(self, silhouette_renderer):
        self.api.call("depth_modality_use_silhouette_checking", self.id, silhouette_renderer.id)

    def data_points(self):
        n = C.c_int()
        cap = self.params.n_points_max
        out = np.zeros(cap, DATA_POINT_DTYPE)
        self.api.call("depth_modality_get_points", self.id, out.ctypes.data_as(C.c_void_p), cap, C.byref(n))
        return out[:min(n.value, cap)]


class Link:
    def __init__(self, api, body=None, parent=None, body2joint_pose=np.eye(4), joint2parent_pose=np.eye(4),
                 free_directions=(1, 1, 1, 1, 1, 1), fixed_body2joint_pose=True):
        self.api = api
        fd = np.asarray(free_directions, np.int32)
        self.id = api.call("link_create", body.id if body is not None else -1,
                           parent.id if parent is not None else -1, fptr(pose_arg(body2joint_pose)),
                           fptr(pose_arg(joint2parent_pose)), iptr(fd), int(fixed_body2joint_pose))
        self._default_body2joint_pose = np.array(body2joint_pose, np.float32)
        self._default_joint2parent_pose = np.array(joint2parent_pose, np.float32)

    def AddModality(self, modality):
        self.api.call("link_add_modality", self.id, modality.id)

    def link2world_pose(self):
        buf = np.zeros(16, np.float32)
        self.api.call("link_get_link2world_pose", self.id, fptr(buf))
        return pose_ret(buf)

    def set_link2world_pose(self, pose):
        """Link::set_link2world_pose (link.cpp:138-140); for a link with a body this is the body's pose"""
        self.api.call("link_set_link2world_pose", self.id, fptr(pose_arg(pose)))

    def ResetJointPoses(self):
        """Link::ResetJointPoses (link.cpp:243-246): back to the poses the link was created with"""
        self.api.call("link_set_joint_poses", self.id, fptr(pose_arg(self._default_body2joint_pose)),
                      fptr(pose_arg(self._default_joint2parent_pose)))

    def set_body2joint_pose(self, pose):
        self.api.call("link_set_joint_poses", self.id, fptr(pose_arg(pose)), None)

    def set_joint2parent_pose(self, pose):
        self.api.call("link_set_joint_poses", self.id, None, fptr(pose_arg(pose)))

    def body2joint_pose(self):
        buf = np.zeros(16, np.float32)
        self.api.call("link_get_joint_poses", self.id, fptr(buf), None)
        return pose_ret(buf)

    def joint2parent_pose(self):
        buf = np.zeros(16, np.float32)
        self.api.call("link_get_joint_poses", self.id, None, fptr(buf))
        return pose_ret(buf)


class Optimizer:
    def __init__(self, api, root_link=None, body=None, modalities=(), tikhonov_parameter_rotation=1000.0,
                 tikhonov_parameter_translation=30000.0):
        self.api = api
        if root_link is not None:
            self.id = api.call("optimizer_create", root_link.id, tikhonov_parameter_rotation,
                               tikhonov_parameter_translation)
        else:
            ids = np.asarray([m.id for m in modalities], np.int32)
            self.id = api.call("optimizer_create_rigid", body.id, len(ids), iptr(ids),
                               tikhonov_parameter_rotation, tikhonov_parameter_translation)


class Constraint:
    def __init__(self, api, optimizer, link1, link2, body12joint1_pose=np.eye(4), body22joint2_pose=np.eye(4),
                 constraint_directions=(0, 0, 0, 0, 0, 0)):
        cd = np.asarray(constraint_directions, np.int32)
        self.id = api.call("constraint_create", optimizer.id, link1.id, link2.id, fptr(pose_arg(body12joint1_pose)),
                           fptr(pose_arg(body22joint2_pose)), iptr(cd))


class SoftConstraint:
    """include/m3t/soft_constraint.h:52-62 (same defaults)"""

    def __init__(self, api, optimizer, link1, link2, body12joint1_pose=np.eye(4), body22joint2_pose=np.eye(4),
                 constraint_directions=(0, 0, 0, 0, 0, 0), max_distance_rotation=0.0, max_distance_translation=0.0,
                 standard_deviation_rotation=0.01, standard_deviation_translation=0.001):
        cd = np.asarray(constraint_directions, np.int32)
        self.id = api.call("soft_constraint_create", optimizer.id, link1.id, link2.id,
                           fptr(pose_arg(body12joint1_pose)), fptr(pose_arg(body22joint2_pose)), iptr(cd),
                           max_distance_rotation, max_distance_translation, standard_deviation_rotation,
                           standard_deviation_translation)


__all__ = ["Tracker", "Body", "ColorCamera", "DepthCamera", "RegionModel", "DepthModel", "RegionModality",
           "DepthModality", "Link", "Optimizer", "Constraint", "SoftConstraint", "RendererGeometry", "ColorHistograms",
           "FocusedBasicDepthRenderer", "FocusedSilhouetteRenderer", "M3TError", "RegionModalityParams",
           "DepthModalityParams"]
