"""ctypes view of the C-ABI declared in include/m3t_hip.h / include/m3t_types.h.

`CApi(path, prefix)` binds every entry point of one shared library.  The
product uses prefix ``m3t_hip_`` (libm3t_hip.so).  The parity tests bind the
CPU oracle (``m3t_oracle_``, oracle/libm3t_oracle.so) through this same class
so both sides are driven by identical host code; the product itself never
loads anything under oracle/.
"""
import ctypes as C

import numpy as np

M3T_MAX_SCALES = 8
M3T_MAX_DISTRIBUTION_LENGTH = 16
M3T_REGION_POINT_FLOATS = 38
M3T_DEPTH_POINT_FLOATS = 36

M3T_OK = 0
M3T_ERR_INVALID_ARGUMENT = -1
M3T_ERR_NOT_SET_UP = -2
M3T_ERR_UNSUPPORTED = -3
M3T_ERR_IO = -4
M3T_ERR_DEVICE = -5
M3T_ERR_NO_MEMORY = -6

c_float_p = C.POINTER(C.c_float)
c_int_p = C.POINTER(C.c_int)


class Intrinsics(C.Structure):
    _fields_ = [("fu", C.c_float), ("fv", C.c_float), ("ppu", C.c_float), ("ppv", C.c_float),
                ("width", C.c_int), ("height", C.c_int)]


class RegionModelDesc(C.Structure):
    _fields_ = [("n_views", C.c_int), ("n_points", C.c_int), ("data_points", c_float_p),
                ("orientations", c_float_p), ("contour_lengths", c_float_p),
                ("stride_depth_offset", C.c_float), ("max_radius_depth_offset", C.c_float)]


class DepthModelDesc(C.Structure):
    _fields_ = [("n_views", C.c_int), ("n_points", C.c_int), ("data_points", c_float_p),
                ("orientations", c_float_p), ("surface_areas", c_float_p),
                ("stride_depth_offset", C.c_float), ("max_radius_depth_offset", C.c_float)]


class BodyGeometry(C.Structure):
    _fields_ = [("vertices", c_float_p), ("n_vertices", C.c_int), ("triangles", c_int_p), ("n_triangles", C.c_int),
                ("geometry2body", C.c_float * 16), ("geometry_counterclockwise", C.c_int),
                ("geometry_enable_culling", C.c_int), ("body_id", C.c_int), ("region_id", C.c_int)]


class ModelGenerationParams(C.Structure):
    """m3t_model_generation_params; defaults = region_model.h:142-148"""
    _fields_ = [("sphere_radius", C.c_float), ("n_divides", C.c_int), ("n_points", C.c_int),
                ("max_radius_depth_offset", C.c_float), ("stride_depth_offset", C.c_float), ("image_size", C.c_int)]

    def __init__(self, sphere_radius=0.8, n_divides=4, n_points=200, max_radius_depth_offset=0.05,
                 stride_depth_offset=0.002, image_size=2000):
        super().__init__(sphere_radius, n_divides, n_points, max_radius_depth_offset, stride_depth_offset, image_size)


class RegionModalityParams(C.Structure):
    """m3t_region_modality_params; defaults = M3T/include/m3t/region_modality.h:411-443."""
    _fields_ = [
        ("n_lines_max", C.c_int), ("use_adaptive_coverage", C.c_int),
        ("reference_contour_length", C.c_float), ("min_continuous_distance", C.c_float),
        ("function_length", C.c_int), ("distribution_length", C.c_int),
        ("function_amplitude", C.c_float), ("function_slope", C.c_float),
        ("learning_rate", C.c_float), ("n_global_iterations", C.c_int),
        ("n_scales", C.c_int), ("scales", C.c_int * M3T_MAX_SCALES),
        ("n_standard_deviations", C.c_int), ("standard_deviations", C.c_float * M3T_MAX_SCALES),
        ("n_histogram_bins", C.c_int), ("learning_rate_f", C.c_float), ("learning_rate_b", C.c_float),
        ("unconsidered_line_length", C.c_float), ("max_considered_line_length", C.c_float),
        ("use_region_checking", C.c_int), ("measure_occlusions", C.c_int),
        ("measured_depth_offset_radius", C.c_float), ("measured_occlusion_radius", C.c_float),
        ("measured_occlusion_threshold", C.c_float), ("model_occlusions", C.c_int),
        ("n_unoccluded_iterations", C.c_int), ("min_n_unoccluded_lines", C.c_int),
        ("modeled_depth_offset_radius", C.c_float), ("modeled_occlusion_radius", C.c_float),
        ("modeled_occlusion_threshold", C.c_float),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.n_lines_max = 200
        self.min_continuous_distance = 3.0
        self.function_length = 8
        self.distribution_length = 12
        self.function_amplitude = 0.43
        self.function_slope = 0.5
        self.learning_rate = 1.3
        self.n_global_iterations = 1
        self.set_scales([6, 4, 2, 1])
        self.set_standard_deviations([15.0, 5.0, 3.5, 1.5])
        self.n_histogram_bins = 16
        self.learning_rate_f = 0.2
        self.learning_rate_b = 0.2
        self.unconsidered_line_length = 0.5
        self.max_considered_line_length = 20.0
        self.measured_depth_offset_radius = 0.01
        self.measured_occlusion_radius = 0.01
        self.measured_occlusion_threshold = 0.03
        self.n_unoccluded_iterations = 10
        self.min_n_unoccluded_lines = 0
        self.modeled_depth_offset_radius = 0.01
        self.modeled_occlusion_radius = 0.01
        self.modeled_occlusion_threshold = 0.03
        for k, v in kw.items():
            if k == "scales":
                self.set_scales(v)
            elif k == "standard_deviations":
                self.set_standard_deviations(v)
            else:
                if not hasattr(self, k):
                    raise AttributeError(k)
                setattr(self, k, v)

    def set_scales(self, v):
        self.n_scales = len(v)
        for i in range(M3T_MAX_SCALES):
            self.scales[i] = int(v[i]) if i < len(v) else 0

    def set_standard_deviations(self, v):
        self.n_standard_deviations = len(v)
        for i in range(M3T_MAX_SCALES):
            self.standard_deviations[i] = float(v[i]) if i < len(v) else 0.0


class DepthModalityParams(C.Structure):
    """m3t_depth_modality_params; defaults = M3T/include/m3t/depth_modality.h:302-321."""
    _fields_ = [
        ("n_points_max", C.c_int), ("use_adaptive_coverage", C.c_int), ("use_depth_scaling", C.c_int),
        ("reference_surface_area", C.c_float), ("stride_length", C.c_float),
        ("n_considered_distances", C.c_int), ("considered_distances", C.c_float * M3T_MAX_SCALES),
        ("n_standard_deviations", C.c_int), ("standard_deviations", C.c_float * M3T_MAX_SCALES),
        ("use_silhouette_checking", C.c_int), ("measure_occlusions", C.c_int),
        ("measured_depth_offset_radius", C.c_float), ("measured_occlusion_radius", C.c_float),
        ("measured_occlusion_threshold", C.c_float), ("model_occlusions", C.c_int),
        ("n_unoccluded_iterations", C.c_int), ("min_n_unoccluded_points", C.c_int),
        ("modeled_depth_offset_radius", C.c_float), ("modeled_occlusion_radius", C.c_float),
        ("modeled_occlusion_threshold", C.c_float),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.n_points_max = 200
        self.stride_length = 0.005
        self.set_considered_distances([0.05, 0.02, 0.01])
        self.set_standard_deviations([0.05, 0.03, 0.02])
        self.measured_depth_offset_radius = 0.01
        self.measured_occlusion_radius = 0.01
        self.measured_occlusion_threshold = 0.03
        self.n_unoccluded_iterations = 10
        self.modeled_depth_offset_radius = 0.01
        self.modeled_occlusion_radius = 0.01
        self.modeled_occlusion_threshold = 0.03
        for k, v in kw.items():
            if k == "considered_distances":
                self.set_considered_distances(v)
            elif k == "standard_deviations":
                self.set_standard_deviations(v)
            else:
                if not hasattr(self, k):
                    raise AttributeError(k)
                setattr(self, k, v)

    def set_considered_distances(self, v):
        self.n_considered_distances = len(v)
        for i in range(M3T_MAX_SCALES):
            self.considered_distances[i] = float(v[i]) if i < len(v) else 0.0

    def set_standard_deviations(self, v):
        self.n_standard_deviations = len(v)
        for i in range(M3T_MAX_SCALES):
            self.standard_deviations[i] = float(v[i]) if i < len(v) else 0.0


class DataLine(C.Structure):
    _fields_ = [("center_f_body", C.c_float * 3), ("center_u", C.c_float), ("center_v", C.c_float),
                ("normal_u", C.c_float), ("normal_v", C.c_float), ("delta_r", C.c_float),
                ("normal_component_to_scale", C.c_float), ("continuous_distance", C.c_float),
                ("mean", C.c_float), ("measured_variance", C.c_float),
                ("distribution", C.c_float * M3T_MAX_DISTRIBUTION_LENGTH),
                ("valid", C.c_int), ("model_point_index", C.c_int)]


class DataPoint(C.Structure):
    _fields_ = [("center_f_body", C.c_float * 3), ("normal_f_body", C.c_float * 3),
                ("center_u", C.c_float), ("center_v", C.c_float), ("depth", C.c_float),
                ("correspondence_center_f_camera", C.c_float * 3),
                ("valid", C.c_int), ("model_point_index", C.c_int)]


DATA_LINE_DTYPE = np.dtype([
    ("center_f_body", np.float32, 3), ("center_u", np.float32), ("center_v", np.float32),
    ("normal_u", np.float32), ("normal_v", np.float32), ("delta_r", np.float32),
    ("normal_component_to_scale", np.float32), ("continuous_distance", np.float32),
    ("mean", np.float32), ("measured_variance", np.float32),
    ("distribution", np.float32, M3T_MAX_DISTRIBUTION_LENGTH),
    ("valid", np.int32), ("model_point_index", np.int32)])
DATA_POINT_DTYPE = np.dtype([
    ("center_f_body", np.float32, 3), ("normal_f_body", np.float32, 3),
    ("center_u", np.float32), ("center_v", np.float32), ("depth", np.float32),
    ("correspondence_center_f_camera", np.float32, 3),
    ("valid", np.int32), ("model_point_index", np.int32)])
assert DATA_LINE_DTYPE.itemsize == C.sizeof(DataLine)
assert DATA_POINT_DTYPE.itemsize == C.sizeof(DataPoint)


class M3TError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("m3t error %d: %s" % (code, message))
        self.code = code


# name -> (restype, argtypes after the context pointer); every one returns int status
# (or a non-negative id) unless noted.
_CTX = C.c_void_p
_SIGNATURES = {
    "region_model_create": [C.POINTER(RegionModelDesc)],
    "region_model_load": [C.c_char_p],
    "depth_model_create": [C.POINTER(DepthModelDesc)],
    "depth_model_load": [C.c_char_p],
    "region_model_info": [C.c_int, c_int_p, c_int_p, c_float_p],
    "depth_model_info": [C.c_int, c_int_p, c_int_p, c_float_p],
    "region_model_closest_view": [C.c_int, c_float_p, c_int_p],
    "depth_model_closest_view": [C.c_int, c_float_p, c_int_p],
    "color_camera_create": [C.POINTER(Intrinsics), c_float_p],
    "depth_camera_create": [C.POINTER(Intrinsics), c_float_p, C.c_float],
    "camera_upload": [C.c_int, C.c_void_p, C.c_size_t],
    "camera_set_world2camera_pose": [C.c_int, c_float_p],
    "body_create": [c_float_p],
    "body_set_body2world_pose": [C.c_int, c_float_p],
    "body_get_body2world_pose": [C.c_int, c_float_p],
    "region_modality_create": [C.POINTER(RegionModalityParams), C.c_int, C.c_int, C.c_int, C.c_int],
    "depth_modality_create": [C.POINTER(DepthModalityParams), C.c_int, C.c_int, C.c_int],
    "link_create": [C.c_int, C.c_int, c_float_p, c_float_p, c_int_p, C.c_int],
    "link_add_modality": [C.c_int, C.c_int],
    "optimizer_create": [C.c_int, C.c_float, C.c_float],
    "optimizer_create_rigid": [C.c_int, C.c_int, c_int_p, C.c_float, C.c_float],
    "constraint_create": [C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_int_p],
    "body_set_geometry": [C.c_int, C.POINTER(BodyGeometry)],
    "renderer_geometry_create": [],
    "renderer_geometry_add_body": [C.c_int, C.c_int],
    "focused_depth_renderer_create": [C.c_int, C.c_int, C.c_int, C.c_float, C.c_float],
    "focused_silhouette_renderer_create": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float],
    "renderer_add_referenced_body": [C.c_int, C.c_int],
    "renderer_start_rendering": [C.c_int],
    "renderer_get_images": [C.c_int, C.POINTER(C.c_uint16), C.POINTER(C.c_uint8), c_float_p, c_int_p],
    "region_modality_model_occlusions": [C.c_int, C.c_int],
    "region_modality_use_region_checking": [C.c_int, C.c_int],
    "depth_modality_model_occlusions": [C.c_int, C.c_int],
    "depth_modality_use_silhouette_checking": [C.c_int, C.c_int],
    "color_histograms_create": [C.c_int, C.c_float, C.c_float],
    "region_modality_use_shared_color_histograms": [C.c_int, C.c_int],
    "refine_poses": [C.c_int, C.c_int],
    "soft_constraint_create": [C.c_int, C.c_int, C.c_int, c_float_p, c_float_p, c_int_p, C.c_float, C.c_float,
                               C.c_float, C.c_float],
    "link_get_link2world_pose": [C.c_int, c_float_p],
    "link_set_link2world_pose": [C.c_int, c_float_p],
    "link_set_joint_poses": [C.c_int, c_float_p, c_float_p],
    "link_get_joint_poses": [C.c_int, c_float_p, c_float_p],
    "calculate_consistent_poses": [],
    "calculate_optimization_begin": [C.POINTER(c_float_p), C.POINTER(C.c_size_t)],
    "calculate_optimization_end": [],
    "tracker_set_iterations": [C.c_int, C.c_int],
    "start_modalities": [C.c_int],
    "calculate_correspondences": [C.c_int, C.c_int],
    "calculate_gradient_and_hessian": [C.c_int, C.c_int, C.c_int],
    "calculate_optimization": [C.c_int, C.c_int, C.c_int],
    "calculate_results": [C.c_int],
    "execute_tracking_step": [C.c_int],
    "execute_tracking_cycle": [C.c_int],
    "sync": [],
    "modality_get_gradient_hessian": [C.c_int, c_float_p, c_float_p],
    "modality_set_gradient_hessian": [C.c_int, c_float_p, c_float_p],
    "modalities_get_gradient_hessian": [c_float_p, C.c_int],
    "region_modality_get_lines": [C.c_int, C.c_void_p, C.c_int, c_int_p],
    "depth_modality_get_points": [C.c_int, C.c_void_p, C.c_int, c_int_p],
    "region_modality_get_histograms": [C.c_int, c_float_p, c_float_p],
    "region_modality_set_histograms": [C.c_int, c_float_p, c_float_p],
}
# entry points only the HIP library has (device plumbing)
_HIP_ONLY = {
    "get_stream": [C.POINTER(C.c_void_p)],
    "region_model_generate": [C.c_int, C.POINTER(ModelGenerationParams)],
    "depth_model_generate": [C.c_int, C.POINTER(ModelGenerationParams)],
    "region_model_generate_associated": [C.c_int, C.POINTER(ModelGenerationParams), C.c_int, c_int_p, c_int_p, c_int_p],
    "depth_model_generate_occluded": [C.c_int, C.POINTER(ModelGenerationParams), C.c_int, c_int_p],
    "region_model_get_views": [C.c_int, c_float_p, c_float_p, c_float_p],
    "depth_model_get_views": [C.c_int, c_float_p, c_float_p, c_float_p],
    "device_info": [C.c_char_p, C.c_size_t, c_int_p, C.POINTER(C.c_size_t)],
    "set_fused_step": [C.c_int],
    "bodies_get_poses": [c_float_p, C.c_int],
    "bodies_set_poses": [c_float_p, C.c_int],
    "camera_set_ring": [C.c_int, C.c_int],
    "camera_upload_slot": [C.c_int, C.c_int, C.c_void_p, C.c_size_t],
    "camera_upload_slot_async": [C.c_int, C.c_int, C.c_void_p, C.c_size_t],
    "host_register": [C.c_void_p, C.c_size_t],
    "host_unregister": [C.c_void_p],
    "ingest_sync": [],
    "camera_slot_sync": [C.c_int, C.c_int],
    "comm_get_unique_id": [C.c_void_p, C.c_size_t],
    "comm_init_rank": [C.c_void_p, C.c_size_t, C.c_int, C.c_int],
    "comm_set": [C.c_void_p],
    "comm_destroy": [],
    "comm_set_reduce_callback": [C.c_void_p, C.c_void_p],
    "calculate_optimization_allreduce": [],
    "comm_get_allreduce_count": [C.POINTER(C.c_longlong)],
    "comm_get_rank_count": [C.POINTER(C.c_int)],
    "cameras_set_ring": [c_int_p, C.c_int, C.c_int],
    "cameras_upload_batch_async": [c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t],
    "set_roi_ingest": [C.c_int, C.c_float],
    "cameras_upload_batch_roi_async": [c_int_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t],
    "roi_get_status": [c_int_p, C.c_int, c_int_p, C.POINTER(C.c_longlong)],
    "roi_get_unrecovered": [c_int_p, C.c_int, c_int_p],
    "reserve_ingest_cus": [C.c_int],
    "camera_select_slot": [C.c_int, C.c_int],
    "cameras_select_slot": [C.c_int],
    "set_kernel_timing": [C.c_int],
    "get_kernel_timing": [c_float_p, c_int_p],
    "get_step_shape": [c_int_p],
    "get_step_kernel": [C.c_char_p, C.c_size_t],
    "debug_log_checksum": [C.c_uint, C.c_uint, C.POINTER(C.c_ulonglong)],
    "set_object_split": [C.c_int],
}


# int fn(void* user, float* device_buffer, size_t count, void* hip_stream): m3t_hip_comm_set_reduce_callback
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


def pose_arg(pose):
    """4x4 numpy pose (row/col indexable) -> float32[16] column-major buffer."""
    p = np.asarray(pose, dtype=np.float32).reshape(4, 4)
    return np.ascontiguousarray(p.T).reshape(16)


def pose_ret(buf):
    return np.array(buf, dtype=np.float32).reshape(4, 4).T.copy()


_NEWER_ENTRY_POINTS = ("comm_set_reduce_callback", "get_step_kernel", "comm_get_allreduce_count", "comm_get_rank_count", "debug_log_checksum", "set_roi_ingest",
                       "cameras_upload_batch_roi_async", "roi_get_status", "roi_get_unrecovered", "reserve_ingest_cus", "camera_slot_sync")


class CApi:
    """One loaded library + one context."""

    def __init__(self, path, prefix, device_id=0):
        self.path = path
        self.prefix = prefix
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        self.is_hip = prefix == "m3t_hip_"
        self._fn = {}
        sigs = dict(_SIGNATURES)
        if self.is_hip:
            sigs.update(_HIP_ONLY)
        for name, args in sigs.items():
            try:
                f = getattr(self.lib, prefix + name)
            except AttributeError:
                if name in _NEWER_ENTRY_POINTS:  # an older build of the library (developer comparisons)
                    continue
                raise
            f.restype = C.c_int
            f.argtypes = [_CTX] + args
            self._fn[name] = f
        create = getattr(self.lib, prefix + "create")
        create.restype = C.c_int
        create.argtypes = [C.POINTER(_CTX), C.c_int]
        self._destroy = getattr(self.lib, prefix + "destroy")
        self._destroy.restype = None
        self._destroy.argtypes = [_CTX]
        self._last_error = getattr(self.lib, prefix + "last_error")
        self._last_error.restype = C.c_char_p
        self._last_error.argtypes = [_CTX]
        self.ctx = _CTX()
        rc = create(C.byref(self.ctx), device_id)
        if rc != 0:
            msg = self._last_error(None)
            raise M3TError(rc, (msg or b"create failed").decode())

    def close(self):
        if self.ctx:
            self._destroy(self.ctx)
            self.ctx = _CTX()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return (self._last_error(self.ctx) or b"").decode()

    def raw(self, name, *args):
        """call without raising; returns the int status"""
        return self._fn[name](self.ctx, *args)

    def call(self, name, *args):
        if name not in self._fn:
            raise M3TError(-3, "%s%s: this entry point exists in the HIP library only" % (self.prefix, name))
        rc = self._fn[name](self.ctx, *args)
        if rc < 0:
            raise M3TError(rc, self.last_error())
        return rc
