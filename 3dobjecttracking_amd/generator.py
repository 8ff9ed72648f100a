"""GenerateConfiguredTracker: the reference's YAML configuration (M3T/include/m3t/generator.h:943-1133) over
the batched device context.

A config file lists, per class name, objects with a `name`, usually a `metafile_path` (relative to the config
file) and references to other objects by name.  Classes of the tracking path are built here — LoaderColorCamera,
LoaderDepthCamera, Body, ColorHistograms, RendererGeometry, FocusedBasicDepthRenderer, FocusedSilhouetteRenderer,
RegionModel, DepthModel, RegionModality, DepthModality, Link, Constraint, SoftConstraint, Optimizer,
StaticDetector, Refiner, Tracker.  Viewers are accepted and ignored (no display on a GPU node); the OpenGL /
sensor / feature-based classes of the reference (TextureModality, ManualDetector, RealSense / AzureKinect
cameras) are refused with the class name in the message.

Errors follow the reference: a missing required parameter or an unknown referenced name raises ValueError with
the reference's message text (generator.h:65-157) where the reference prints it and returns false.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import config as cfg
from . import host
from ._capi import DepthModalityParams, RegionModalityParams

_IGNORED = ("ImageColorViewer", "ImageDepthViewer", "NormalColorViewer", "NormalDepthViewer")
_REFUSED = ("TextureModality", "ManualDetector", "RealSenseColorCamera", "RealSenseDepthCamera",
            "AzureKinectColorCamera", "AzureKinectDepthCamera")


# ---------------------------------------------------------------------------------------------------------
# objects with metafiles
# ---------------------------------------------------------------------------------------------------------
class _LoaderCamera:
    """LoaderColorCamera / LoaderDepthCamera (loader_camera.cpp): images named
    <pre><zero-padded load_index><post>.<type> in load_directory, one per UpdateImage()."""

    def _init_loader(self, load_directory, image_name_pre, load_index, n_leading_zeros, image_name_post,
                     load_image_type):
        self.load_directory = load_directory
        self.image_name_pre, self.image_name_post = image_name_pre, image_name_post
        self.load_index, self.n_leading_zeros = int(load_index), int(n_leading_zeros)
        self.load_image_type = load_image_type

    @property
    def image(self):
        """Camera::image() (camera.h): the frame of the last UpdateImage.  With prefetch on, the pixels live in a
        page-locked slab that a worker thread overwrites two frames later: callers get their own copy, valid as long
        as they keep it (viewers, recorders)."""
        slab = getattr(self, "_image_slab", None)
        return slab.copy() if slab is not None else getattr(self, "_image", None)

    @image.setter
    def image(self, value):
        self._image_slab = None
        self._image = value

    def image_path(self):
        s = str(self.load_index)
        n_zeros = max(self.n_leading_zeros - len(s), 0)  # loader_camera.cpp:83-88
        return os.path.join(self.load_directory, self.image_name_pre + "0" * n_zeros + s + self.image_name_post +
                            "." + self.load_image_type)

    def set_load_index(self, load_index):
        self.load_index = int(load_index)
        self._drop_pipeline()

    # ---- overlapped ingest (what replaces loader_camera.cpp:76-98's blocking imread + upload) ----
    # Sequences are read in order, so while the tracker works on frame k a worker thread decodes frame k + 2 into a
    # page-locked slab and the main thread has frame k + 1 crossing PCIe into the next slot of a three-slot device
    # ring (m3t_hip_camera_upload_slot_async on the library's copy stream).  UpdateImage(k) then is a pointer switch.
    # HIP contexts only; the arithmetic of the step never sees the difference (same pixels, same order).
    _PIPELINE_SLOTS = 3

    def enable_prefetch(self, enable=True):
        """decode + upload ahead of UpdateImage (default for HIP contexts; switch off to get the blocking reference
        behaviour, e.g. when frames are written while they are being tracked)"""
        self._prefetch = bool(enable) and getattr(self.api, "is_hip", False)
        self._drop_pipeline()

    def enable_roi_ingest(self, enable=True, margin_px=24.0, adaptive=False, reserve_cus=0):
        """ROI ingest for this camera's pipeline (m3t_hip_set_roi_ingest: a setting of the whole context, every
        loader camera of a tracker should be given the same one): of frame k + 1 only the rectangle the trackers can
        read crosses PCIe, pulled by a kernel out of the page-locked slab while frame k is tracked; a body that
        outruns its rectangle has its step repeated on the whole frame by the library, so the poses are those of
        whole frames bit for bit.  adaptive: per-body margins from the motion over the last step (at most margin_px).
        reserve_cus (a multiple of 32): compute units kept free of the tracking kernels for the pull
        (m3t_hip_reserve_ingest_cus).  HIP contexts with prefetch on; otherwise ignored."""
        self._roi = bool(enable) and getattr(self.api, "is_hip", False)
        self._drop_pipeline()
        if getattr(self.api, "is_hip", False):
            self.api.call("set_roi_ingest", (2 if adaptive else 1) if enable else 0, C.c_float(margin_px))
            if reserve_cus or not enable:
                self.api.call("reserve_ingest_cus", int(reserve_cus) if enable else 0)

    def _drop_pipeline(self):
        for job in getattr(self, "_jobs", {}).values():
            job["thread"].join()
        if getattr(self, "_uploaded", None):
            self.api.call("ingest_sync")  # the slabs must not be reused while a copy reads them
        self._jobs, self._uploaded = {}, {}

    def release(self):
        """before the camera (or its arrays) go away: no copy in flight, the slabs no longer page-locked"""
        self._drop_pipeline()
        if hasattr(self, "_slabs"):
            import ctypes as C
            for a in self._slabs:
                try:
                    self.api.call("host_unregister", a.ctypes.data_as(C.c_void_p))
                except Exception:  # (the context may be gone already: it unregisters what is left itself)
                    pass
            del self._slabs

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass

    def _slab(self, slot):
        if not hasattr(self, "_slabs"):
            import ctypes as C
            shape = (self.height, self.width, 3) if self.bpp == 3 else (self.height, self.width)
            self._slabs = [np.zeros(shape, np.uint8 if self.bpp == 3 else np.uint16)
                           for _ in range(self._PIPELINE_SLOTS)]
            for a in self._slabs:
                self.api.call("host_register", a.ctypes.data_as(C.c_void_p), a.nbytes)
            # (a ring of the batch kind, with this camera as its only member: what the rectangle upload needs)
            self.api.call("cameras_set_ring", (C.c_int * 1)(self.id), 1, self._PIPELINE_SLOTS)
        return self._slabs[slot]

    def _start_decode(self, index):
        """worker thread: image `index` into its slab (no device call: the C-ABI belongs to the main thread)"""
        import threading
        if index in self._jobs or index in self._uploaded:
            return
        slab = self._slab(index % self._PIPELINE_SLOTS)
        # the copy that last read this slab must have left it (a no-op in steady state: UpdateImage has waited for it;
        # after set_load_index / enable_prefetch a popped upload may still be in flight)
        self.slot_sync(index % self._PIPELINE_SLOTS)
        saved, self.load_index = self.load_index, index
        path = self.image_path()
        self.load_index = saved
        job = {"ok": False, "path": path, "error": None}

        def work():
            try:
                np.copyto(slab, self._decode(path))
                job["ok"] = True
            except (OSError, ValueError) as e:
                job["error"] = e

        job["thread"] = threading.Thread(target=work, daemon=True)
        job["thread"].start()
        self._jobs[index] = job

    def _upload_when_decoded(self, index, wait):
        """main thread: hand a decoded slab to the copy stream; returns False if it is not there (yet)"""
        if index in self._uploaded:
            return True
        job = self._jobs.get(index)
        if job is None:
            return False
        if not wait and job["thread"].is_alive():
            return False
        job["thread"].join()
        del self._jobs[index]
        if not job["ok"]:
            self._failed = (job["path"], job["error"])
            return False
        slot = index % self._PIPELINE_SLOTS
        slab = self._slabs[slot]
        if getattr(self, "_roi", False):
            # the trackers' rectangle only (whole frames until a fused step has run, and whenever rectangles are not
            # possible: the library decides); the slab stays untouched until slot_sync
            self.api.call("cameras_upload_batch_roi_async", (C.c_int * 1)(self.id), 1, slot,
                          slab.ctypes.data_as(C.c_void_p), slab.nbytes, slab.strides[0])
        else:
            self.upload_slot(slot, slab, asynchronous=True)
        self._uploaded[index] = slot
        return True

    def UpdateImage(self, synchronized=True):
        if getattr(self, "_prefetch", None) is None:
            self.enable_prefetch(True)
        if not self._prefetch:
            path = self.image_path()
            try:
                image = self._decode(path)
            except (OSError, ValueError) as e:
                sys.stderr.write("Could not read image from %s (%s)\n" % (path, e))
                return False
            self._image_slab = None
            self._image = image
            self.load_index += 1
            return super().UpdateImage(image)
        k = self.load_index
        self._failed = None
        self._start_decode(k)  # (already running or done unless this is the first frame / the index was set)
        if not self._upload_when_decoded(k, wait=True):
            path, e = self._failed or (self.image_path(), "not decoded")
            sys.stderr.write("Could not read image from %s (%s)\n" % (path, e))
            self._drop_pipeline()
            return False
        slot = self._uploaded.pop(k)
        # the slab frame k + 2 is about to be decoded into is the one frame k - 1 was copied out of, three uploads of
        # THIS camera ago: wait for that copy and for no other (the other cameras' copies of frame k + 1, enqueued a
        # moment ago by their UpdateImage, stay in flight next to the coming step)
        self.slot_sync((k + 2) % self._PIPELINE_SLOTS)
        self._image_slab = self._slabs[slot]  # (`image` hands out a copy: the slab is decoded into again two frames on)
        self.select_slot(slot)  # the step that follows waits for this slot's copy, and only for it
        self.load_index += 1
        # frame k + 1: decoded while the previous step ran -> its copy overlaps the coming step; frame k + 2: decode now
        self._start_decode(k + 1)
        self._upload_when_decoded(k + 1, wait=False)
        # (slot (k + 2) % 3 is free: frame k - 1 was read by a step the main thread has already enqueued; the library
        # orders an asynchronous upload behind the last step that read its slot)
        self._start_decode(k + 2)
        return True


def _loader_meta(path):
    d = cfg.read_yaml(path)
    cfg.required(d, ("load_directory", "intrinsics"), "body", path)  # (the reference's message says "body")
    i = d["intrinsics"]
    cfg.required(i, ("f_u", "f_v", "pp_x", "pp_y", "width", "height"), "intrinsics", path)
    c2w = cfg.pose(d["camera2world_pose"]) if "camera2world_pose" in d else np.eye(4, dtype=np.float32)
    return d, i, c2w, cfg.relative_to(path, d["load_directory"])


def _inverse_pose(p):
    """Transform3fA::inverse() (Affine: general 3x3 inverse), world2camera_pose_ of loader_camera.cpp:145"""
    return np.linalg.inv(np.asarray(p, np.float64)).astype(np.float32)


class LoaderColorCamera(_LoaderCamera, host.ColorCamera):
    def __init__(self, api, load_directory, intrinsics, image_name_pre="", load_index=0, n_leading_zeros=0,
                 image_name_post="", load_image_type="png", camera2world_pose=np.eye(4)):
        fu, fv, ppu, ppv, w, h = intrinsics
        host.ColorCamera.__init__(self, api, fu, fv, ppu, ppv, int(w), int(h), _inverse_pose(camera2world_pose))
        self._init_loader(load_directory, image_name_pre, load_index, n_leading_zeros, image_name_post,
                          load_image_type)

    @classmethod
    def from_metafile(cls, api, path):
        d, i, c2w, directory = _loader_meta(path)
        return cls(api, directory, (i["f_u"], i["f_v"], i["pp_x"], i["pp_y"], i["width"], i["height"]),
                   d.get("image_name_pre", ""), d.get("load_index", 0), d.get("n_leading_zeros", 0),
                   d.get("image_name_post", ""), d.get("load_image_type", "png"), c2w)

    @staticmethod
    def _decode(path):
        """cv::imread(IMREAD_UNCHANGED) of a colour image: 8-bit B, G, R"""
        from PIL import Image
        im = Image.open(path)
        if im.mode not in ("RGB", "RGBA", "P"):
            raise ValueError("not a colour image: mode %s" % im.mode)
        return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])


class LoaderDepthCamera(_LoaderCamera, host.DepthCamera):
    def __init__(self, api, load_directory, intrinsics, depth_scale, image_name_pre="", load_index=0,
                 n_leading_zeros=0, image_name_post="", load_image_type="png", camera2world_pose=np.eye(4)):
        fu, fv, ppu, ppv, w, h = intrinsics
        host.DepthCamera.__init__(self, api, fu, fv, ppu, ppv, int(w), int(h), float(depth_scale),
                                  _inverse_pose(camera2world_pose))
        self._init_loader(load_directory, image_name_pre, load_index, n_leading_zeros, image_name_post,
                          load_image_type)

    @classmethod
    def from_metafile(cls, api, path):
        d, i, c2w, directory = _loader_meta(path)
        cfg.required(d, ("depth_scale",), "body", path)
        return cls(api, directory, (i["f_u"], i["f_v"], i["pp_x"], i["pp_y"], i["width"], i["height"]),
                   d["depth_scale"], d.get("image_name_pre", ""), d.get("load_index", 0),
                   d.get("n_leading_zeros", 0), d.get("image_name_post", ""), d.get("load_image_type", "png"), c2w)

    @staticmethod
    def _decode(path):
        """cv::imread(IMREAD_UNCHANGED) of a 16-bit single-channel image"""
        from PIL import Image
        a = np.asarray(Image.open(path))
        if a.ndim != 2:
            raise ValueError("not a single-channel depth image")
        return np.ascontiguousarray(a.astype(np.uint16))


_next_body_id = [1]  # body.cpp:11


class Body(host.Body):
    """m3t::Body with its mesh (body.cpp:13-42,152-252)"""

    def __init__(self, api, name, geometry_path, geometry_unit_in_meter, geometry_counterclockwise,
                 geometry_enable_culling, geometry2body_pose, body_id=None, region_id=None):
        host.Body.__init__(self, api)
        self.name = name
        if body_id is None:
            body_id = _next_body_id[0]
            _next_body_id[0] = (_next_body_id[0] + 1) % 256
        self.body_id = int(body_id)
        self.region_id = int(region_id) if region_id is not None else self.body_id
        self.geometry_path = str(geometry_path)
        self.geometry_unit_in_meter = float(geometry_unit_in_meter)
        self.geometry_counterclockwise = bool(geometry_counterclockwise)
        self.geometry_enable_culling = bool(geometry_enable_culling)
        self.geometry2body_pose = np.asarray(geometry2body_pose, np.float32).reshape(4, 4)
        self.vertices, self.triangles = cfg.load_obj(self.geometry_path, self.geometry_unit_in_meter)
        g = self.geometry2body_pose
        moved = self.vertices @ g[:3, :3].T + g[:3, 3]
        self.maximum_body_diameter = cfg.maximum_body_diameter(moved)
        self.set_geometry(self.vertices, self.triangles, self.geometry2body_pose, self.geometry_counterclockwise,
                          self.geometry_enable_culling, self.body_id, self.region_id)

    @classmethod
    def from_metafile(cls, api, name, path):
        d = cfg.read_yaml(path)
        cfg.required(d, ("geometry_path", "geometry_unit_in_meter", "geometry_counterclockwise",
                         "geometry_enable_culling", "geometry2body_pose"), "body", path)
        gp = d["geometry_path"]
        gp = os.path.join(os.path.dirname(os.path.abspath(path)), name + ".obj") if gp == "INFER_FROM_NAME" \
            else cfg.relative_to(path, gp)
        return cls(api, name, gp, d["geometry_unit_in_meter"], d["geometry_counterclockwise"],
                   d["geometry_enable_culling"], cfg.pose(d["geometry2body_pose"]), d.get("body_id"),
                   d.get("region_id"))

    def body_data(self):
        return cfg.BodyData(self.geometry_path, self.geometry_unit_in_meter, self.geometry_counterclockwise,
                            self.geometry_enable_culling, self.maximum_body_diameter, self.geometry2body_pose)


_MODEL_DEFAULTS = dict(sphere_radius=0.8, n_divides=4, n_points=200, max_radius_depth_offset=0.05,
                       stride_depth_offset=0.002, use_random_seed=False, image_size=2000)  # model.h:132-138


def _model_from_metafile(api, name, path, body, region):
    """Model::SetUp (region_model.cpp:28-56, depth_model.cpp:28-56): load model_path if it was generated with
    these parameters for this body, else generate and save it."""
    d = cfg.read_yaml(path)
    cfg.required(d, ("model_path",), "body", path)
    params = dict(_MODEL_DEFAULTS)
    for k in params:
        if k in d:
            params[k] = type(params[k])(d[k])
    mp = d["model_path"]
    mp = os.path.join(os.path.dirname(os.path.abspath(path)), name + ".bin") if mp == "INFER_FROM_NAME" \
        else cfg.relative_to(path, mp)
    if params["use_random_seed"]:
        raise ValueError("use_random_seed: models are generated with the fixed seed only")
    klass = host.RegionModel if region else host.DepthModel
    if cfg.model_bin_matches(mp, region, params, body.body_data()):
        model = klass(api, path=mp)
    else:
        gen = {k: params[k] for k in ("sphere_radius", "n_divides", "n_points", "max_radius_depth_offset",
                                      "stride_depth_offset", "image_size")}
        model = klass.generate(api, body, **gen)
        pts, ori, ext = model.views()
        try:
            cfg.write_model_bin(mp, region, params, body.body_data(), pts, ori, ext)
        except OSError as e:  # a read-only data directory does not stop tracking
            sys.stderr.write("Could not save model file %s (%s)\n" % (mp, e))
    model.name, model.model_path, model.parameters = name, mp, params
    return model


class StaticDetector:
    """static_detector.cpp + Detector::UpdatePoses (detector.cpp:42-53)"""

    def __init__(self, name, optimizer, link2world_pose, reset_joint_poses=True):
        self.name, self.optimizer = name, optimizer
        self.link2world_pose = np.asarray(link2world_pose, np.float32).reshape(4, 4)
        self.reset_joint_poses = bool(reset_joint_poses)

    @classmethod
    def from_metafile(cls, name, path, optimizer):
        d = cfg.read_yaml(path)
        cfg.required(d, ("link2world_pose",), "static detector", path)
        return cls(name, optimizer, cfg.pose(d["link2world_pose"]), d.get("reset_joint_poses", True))

    def DetectPoses(self, names, detected_names=None):
        if self.optimizer.name in names:
            self.optimizer.root_link.set_link2world_pose(self.link2world_pose)
            if self.reset_joint_poses:
                for link in self.optimizer.ReferencedLinks():
                    link.ResetJointPoses()
            if detected_names is not None:
                detected_names.add(self.optimizer.name)
        return True


# ---------------------------------------------------------------------------------------------------------
# the generated tracker
# ---------------------------------------------------------------------------------------------------------
class GeneratedTracker(host.Tracker):
    """m3t::Tracker as GenerateConfiguredTracker returns it: the step methods of host.Tracker plus the named
    objects, the cameras' UpdateImage and the detectors (tracker.cpp:209-330)."""

    def __init__(self, api, name, n_corr_iterations=5, n_update_iterations=2):
        host.Tracker.__init__(self, api, n_corr_iterations, n_update_iterations)
        self.name = name
        self.objects = {}  # class name -> {name: object}
        self.optimizers, self.detectors, self.refiners, self.cameras, self.bodies = [], [], [], [], []
        self.start_renderers = []
        self.set_up = False

    def SetUp(self, set_up_all_objects=True):
        """every object was set up when it was built; the cameras load their first image here (Camera::SetUp)"""
        for cam in self.cameras:
            if not cam.UpdateImage(True):
                return False
        self.set_up = True
        return True

    def enable_roi_ingest(self, enable=True, margin_px=24.0, adaptive=False, reserve_cus=0):
        """ROI ingest for every loader camera of the tracker (_LoaderCamera.enable_roi_ingest): of each frame only
        the trackers' rectangle crosses PCIe; poses equal those of whole frames bit for bit (a body that outruns its
        rectangle is repeated on the whole frame inside ExecuteTrackingStep)"""
        for cam in self.cameras:
            if hasattr(cam, "enable_roi_ingest"):
                cam.enable_roi_ingest(enable, margin_px, adaptive, reserve_cus)

    def body_ptrs(self):
        return list(self.bodies)

    def UpdateCameras(self, iteration):
        if not self.set_up:
            sys.stderr.write("Set up tracker %s first\n" % self.name)
            return False
        return all(cam.UpdateImage(True) for cam in self.cameras)

    def DetectPoses(self, names, detected_names=None):
        if not self.set_up:
            sys.stderr.write("Set up tracker %s first\n" % self.name)
            return False
        ok = all(d.DetectPoses(set(names), detected_names) for d in self.detectors)
        return ok and self.CalculateConsistentPoses()

    def RefinePoses(self, names=None):
        r = self.refiners[0] if self.refiners else dict(n_corr_iterations=7, n_update_iterations=2)
        return host.Tracker.RefinePoses(self, r["n_corr_iterations"], r["n_update_iterations"])

    def RunTrackerProcess(self, n_frames, names=None):
        """the loop of Tracker::RunTrackerProcess (tracker.cpp:257-330) without viewers and keyboard: detect and
        start once, then UpdateCameras + ExecuteTrackingStep per frame.  Returns the number of frames tracked."""
        if not self.set_up:
            sys.stderr.write("Set up tracker %s first\n" % self.name)
            return False
        names = set(names) if names is not None else {o.name for o in self.optimizers}
        if not (self.DetectPoses(names) and self.StartModalities(0)):
            return False
        for iteration in range(n_frames):
            if iteration > 0 and not self.UpdateCameras(iteration):
                return iteration
            if not self.ExecuteTrackingStep(iteration):
                return iteration
        return n_frames


def _entries(d, class_name, required, path):
    out = []
    for node in d.get(class_name) or []:
        missing = [k for k in required if k not in node]
        if missing:
            raise ValueError('Required parameter "%s" was not found for class %s in %s' % (missing[0], class_name, path))
        out.append(node)
    return out


def _get(objects, class_name, node, key, *kinds):
    name = node[key]
    for kind in kinds:
        if name in objects.get(kind, {}):
            return objects[kind][name]
    raise ValueError("Object %s required by %s %s was not found" % (name, class_name, node.get("name", "")))


def _modality_params(klass, node, configfile_path, allowed):
    kw = {}
    if "metafile_path" in node:
        meta = cfg.read_yaml(cfg.relative_to(configfile_path, node["metafile_path"]))
        for k in allowed:
            if k in meta:
                kw[k] = meta[k]
    return klass(**kw)


_REGION_KEYS = ("n_lines_max", "use_adaptive_coverage", "reference_contour_length", "min_continuous_distance",
                "function_length", "distribution_length", "function_amplitude", "function_slope", "learning_rate",
                "n_global_iterations", "scales", "standard_deviations", "n_histogram_bins", "learning_rate_f",
                "learning_rate_b", "unconsidered_line_length", "max_considered_line_length",
                "measured_depth_offset_radius", "measured_occlusion_radius", "measured_occlusion_threshold",
                "modeled_depth_offset_radius", "modeled_occlusion_radius", "modeled_occlusion_threshold",
                "n_unoccluded_iterations", "min_n_unoccluded_lines")  # region_modality.cpp:810-850
_DEPTH_KEYS = ("n_points_max", "use_adaptive_coverage", "use_depth_scaling", "reference_surface_area",
               "stride_length", "considered_distances", "standard_deviations", "measure_occlusions",
               "measured_depth_offset_radius", "measured_occlusion_radius", "measured_occlusion_threshold",
               "modeled_depth_offset_radius", "modeled_occlusion_radius", "modeled_occlusion_threshold",
               "n_unoccluded_iterations", "min_n_unoccluded_points")  # depth_modality.cpp:560-600


def GenerateConfiguredTracker(api, configfile_path):
    """generator.h:943-1133.  `api` is an open device context (3dobjecttracking_amd.open_context); returns the
    GeneratedTracker, whose .objects[class_name][name] holds everything that was configured."""
    path = str(configfile_path)
    d = cfg.read_yaml(path)
    for class_name in _REFUSED:
        if d.get(class_name):
            raise ValueError("Class %s of %s is outside the tracking path this library replaces" % (class_name, path))
    ob = {}

    def put(class_name, node, obj):
        obj.name = node["name"]
        ob.setdefault(class_name, {})[node["name"]] = obj
        return obj

    def meta(node):
        return cfg.relative_to(path, node["metafile_path"])

    for node in _entries(d, "Body", ("name", "metafile_path"), path):
        put("Body", node, Body.from_metafile(api, node["name"], meta(node)))
    for node in _entries(d, "ColorHistograms", ("name",), path):
        m = cfg.read_yaml(meta(node)) if "metafile_path" in node else {}
        put("ColorHistograms", node, host.ColorHistograms(api, int(m.get("n_bins", 16)),
                                                          float(m.get("learning_rate_f", 0.2)),
                                                          float(m.get("learning_rate_b", 0.2))))
    for node in _entries(d, "RendererGeometry", ("name", "bodies"), path):
        rg = put("RendererGeometry", node, host.RendererGeometry(api))
        for name in node["bodies"]:
            rg.AddBody(_get(ob, "RendererGeometry", {"name": node["name"], "b": name}, "b", "Body"))
    for node in _entries(d, "LoaderColorCamera", ("name", "metafile_path"), path):
        put("LoaderColorCamera", node, LoaderColorCamera.from_metafile(api, meta(node)))
    for node in _entries(d, "LoaderDepthCamera", ("name", "metafile_path"), path):
        put("LoaderDepthCamera", node, LoaderDepthCamera.from_metafile(api, meta(node)))
    cams = ("LoaderColorCamera", "LoaderDepthCamera")
    for class_name, silhouette in (("FocusedBasicDepthRenderer", False), ("FocusedSilhouetteRenderer", True)):
        for node in _entries(d, class_name, ("name", "renderer_geometry", "camera", "referenced_bodies"), path):
            m = cfg.read_yaml(meta(node)) if "metafile_path" in node else {}
            rg = _get(ob, class_name, node, "renderer_geometry", "RendererGeometry")
            cam = _get(ob, class_name, node, "camera", *cams)
            kw = dict(image_size=int(m.get("image_size", 200)), z_min=float(m.get("z_min", 0.02)),
                      z_max=float(m.get("z_max", 10.0)))
            r = host.FocusedSilhouetteRenderer(api, rg, cam, int(m.get("id_type", 0)), **kw) if silhouette \
                else host.FocusedBasicDepthRenderer(api, rg, cam, **kw)
            put(class_name, node, r)
            for name in node["referenced_bodies"]:
                r.AddReferencedBody(_get(ob, class_name, {"name": node["name"], "b": name}, "b", "Body"))
    for class_name, region in (("RegionModel", True), ("DepthModel", False)):
        for node in _entries(d, class_name, ("name", "metafile_path", "body"), path):
            for k in ("fixed_bodies", "movable_bodies", "fixed_same_region_bodies", "movable_same_region_bodies",
                      "occlusion_bodies"):
                if node.get(k):
                    raise ValueError("%s %s: associated bodies (%s) are not generated by this library" %
                                     (class_name, node["name"], k))
            body = _get(ob, class_name, node, "body", "Body")
            put(class_name, node, _model_from_metafile(api, node["name"], meta(node), body, region))
    modalities = {}
    for node in _entries(d, "RegionModality", ("name", "body", "color_camera", "region_model"), path):
        params = _modality_params(RegionModalityParams, node, path, _REGION_KEYS)
        depth_camera = None
        if node.get("measure_occlusions"):
            depth_camera = _get(ob, "RegionModality", node["measure_occlusions"] | {"name": node["name"]},
                                "depth_camera", "LoaderDepthCamera")
            params.measure_occlusions = 1
        mod = host.RegionModality(api, _get(ob, "RegionModality", node, "body", "Body"),
                                  _get(ob, "RegionModality", node, "color_camera", "LoaderColorCamera"),
                                  _get(ob, "RegionModality", node, "region_model", "RegionModel"), depth_camera, params)
        if node.get("model_occlusions"):
            mod.ModelOcclusions(_get(ob, "RegionModality", node["model_occlusions"] | {"name": node["name"]},
                                     "focused_depth_renderer", "FocusedBasicDepthRenderer",
                                     "FocusedSilhouetteRenderer"))
        if node.get("use_region_checking"):
            mod.UseRegionChecking(_get(ob, "RegionModality", node["use_region_checking"] | {"name": node["name"]},
                                       "focused_silhouette_renderer", "FocusedSilhouetteRenderer"))
        if node.get("use_shared_color_histograms"):
            mod.UseSharedColorHistograms(_get(ob, "RegionModality",
                                              node["use_shared_color_histograms"] | {"name": node["name"]},
                                              "color_histograms", "ColorHistograms"))
        modalities[node["name"]] = put("RegionModality", node, mod)
    for node in _entries(d, "DepthModality", ("name", "body", "depth_camera", "depth_model"), path):
        params = _modality_params(DepthModalityParams, node, path, _DEPTH_KEYS)
        mod = host.DepthModality(api, _get(ob, "DepthModality", node, "body", "Body"),
                                 _get(ob, "DepthModality", node, "depth_camera", "LoaderDepthCamera"),
                                 _get(ob, "DepthModality", node, "depth_model", "DepthModel"), params)
        if node.get("model_occlusions"):
            mod.ModelOcclusions(_get(ob, "DepthModality", node["model_occlusions"] | {"name": node["name"]},
                                     "focused_depth_renderer", "FocusedBasicDepthRenderer",
                                     "FocusedSilhouetteRenderer"))
        if node.get("use_silhouette_checking"):
            mod.UseSilhouetteChecking(_get(ob, "DepthModality",
                                           node["use_silhouette_checking"] | {"name": node["name"]},
                                           "focused_silhouette_renderer", "FocusedSilhouetteRenderer"))
        modalities[node["name"]] = put("DepthModality", node, mod)

    # links: parents before children (generator.h:587-641 wires child_links in a second pass)
    link_nodes = {n["name"]: n for n in _entries(d, "Link", ("name",), path)}
    parent_of = {}
    for n in link_nodes.values():
        for child in n.get("child_links") or []:
            if child not in link_nodes:
                raise ValueError("Object %s required by Link %s was not found" % (child, n["name"]))
            parent_of[child] = n["name"]

    def build_link(name, seen=()):
        if name in ob.get("Link", {}):
            return ob["Link"][name]
        if name in seen:
            raise ValueError("Link %s is its own ancestor" % name)
        node = link_nodes[name]
        parent = build_link(parent_of[name], seen + (name,)) if name in parent_of else None
        m = cfg.read_yaml(meta(node)) if "metafile_path" in node else {}
        body = _get(ob, "Link", node, "body", "Body") if node.get("body") else None
        b2j = cfg.pose(m["body2joint_pose"]) if "body2joint_pose" in m else np.eye(4, dtype=np.float32)
        j2p = cfg.pose(m["joint2parent_pose"]) if "joint2parent_pose" in m else np.eye(4, dtype=np.float32)
        link = host.Link(api, body, parent, b2j, j2p, [int(bool(x)) for x in m.get("free_directions", [1] * 6)],
                         bool(m.get("fixed_body2joint_pose", True)))
        link.default_body2joint_pose, link.default_joint2parent_pose = b2j, j2p
        link.body, link.parent, link.children = body, parent, []
        if parent is not None:
            parent.children.append(link)
        if "link2world_pose" in m:
            link.set_link2world_pose(cfg.pose(m["link2world_pose"]))
        for mod_name in node.get("modalities") or []:
            if mod_name not in modalities:
                raise ValueError("Object %s required by Link %s was not found" % (mod_name, name))
            link.AddModality(modalities[mod_name])
        return put("Link", node, link)

    for name in link_nodes:
        build_link(name)

    constraint_nodes = {n["name"]: n for n in _entries(d, "Constraint", ("name", "link1", "link2"), path)}
    soft_nodes = {n["name"]: n for n in _entries(d, "SoftConstraint", ("name", "link1", "link2"), path)}
    tracker = None
    optimizers = {}
    for node in _entries(d, "Optimizer", ("name", "root_link"), path):
        m = cfg.read_yaml(meta(node)) if "metafile_path" in node else {}
        root = _get(ob, "Optimizer", node, "root_link", "Link")
        opt = host.Optimizer(api, root, tikhonov_parameter_rotation=float(m.get("tikhonov_parameter_rotation", 1000.0)),
                             tikhonov_parameter_translation=float(m.get("tikhonov_parameter_translation", 30000.0)))
        opt.root_link = root
        opt.ReferencedLinks = lambda root=root: _subtree(root)
        for cname in node.get("constraints") or []:
            if cname not in constraint_nodes:
                raise ValueError("Object %s required by Optimizer %s was not found" % (cname, node["name"]))
            cn = constraint_nodes[cname]
            cm = cfg.read_yaml(meta(cn)) if "metafile_path" in cn else {}
            put("Constraint", cn, host.Constraint(
                api, opt, _get(ob, "Constraint", cn, "link1", "Link"), _get(ob, "Constraint", cn, "link2", "Link"),
                cfg.pose(cm["body12joint1_pose"]) if "body12joint1_pose" in cm else np.eye(4),
                cfg.pose(cm["body22joint2_pose"]) if "body22joint2_pose" in cm else np.eye(4),
                [int(bool(x)) for x in cm.get("constraint_directions", [0] * 6)]))
        for cname in node.get("soft_constraints") or []:
            if cname not in soft_nodes:
                raise ValueError("Object %s required by Optimizer %s was not found" % (cname, node["name"]))
            cn = soft_nodes[cname]
            cm = cfg.read_yaml(meta(cn)) if "metafile_path" in cn else {}
            put("SoftConstraint", cn, host.SoftConstraint(
                api, opt, _get(ob, "SoftConstraint", cn, "link1", "Link"),
                _get(ob, "SoftConstraint", cn, "link2", "Link"),
                cfg.pose(cm["body12joint1_pose"]) if "body12joint1_pose" in cm else np.eye(4),
                cfg.pose(cm["body22joint2_pose"]) if "body22joint2_pose" in cm else np.eye(4),
                [int(bool(x)) for x in cm.get("constraint_directions", [0] * 6)],
                float(cm.get("max_distance_rotation", 0.0)), float(cm.get("max_distance_translation", 0.0)),
                float(cm.get("standard_deviation_rotation", 0.01)),
                float(cm.get("standard_deviation_translation", 0.001))))
        optimizers[node["name"]] = put("Optimizer", node, opt)

    for node in _entries(d, "StaticDetector", ("name", "metafile_path", "optimizer"), path):
        put("StaticDetector", node, StaticDetector.from_metafile(node["name"], meta(node),
                                                                 _get(ob, "StaticDetector", node, "optimizer",
                                                                      "Optimizer")))
    for node in _entries(d, "Refiner", ("name", "optimizers"), path):
        m = cfg.read_yaml(meta(node)) if "metafile_path" in node else {}
        ob.setdefault("Refiner", {})[node["name"]] = dict(n_corr_iterations=int(m.get("n_corr_iterations", 7)),
                                                          n_update_iterations=int(m.get("n_update_iterations", 2)))

    trackers = _entries(d, "Tracker", ("name", "optimizers"), path)
    if len(trackers) < 1:
        raise ValueError("No tracker was configured in %s" % path)
    if len(trackers) > 1:
        raise ValueError("More than one tracker was configured in %s" % path)
    node = trackers[0]
    m = cfg.read_yaml(meta(node)) if "metafile_path" in node else {}
    tracker = GeneratedTracker(api, node["name"], int(m.get("n_corr_iterations", 5)),
                               int(m.get("n_update_iterations", 2)))
    for name in node["optimizers"]:
        if name not in optimizers:
            raise ValueError("Object %s required by Tracker %s was not found" % (name, node["name"]))
        tracker.optimizers.append(optimizers[name])
    unused = set(optimizers) - set(node["optimizers"])
    if unused:
        raise ValueError("Optimizers %s are configured but not part of tracker %s: one device context runs one "
                         "tracker" % (sorted(unused), node["name"]))
    for name in node.get("detectors") or []:
        tracker.detectors.append(_get(ob, "Tracker", {"name": node["name"], "d": name}, "d", "StaticDetector"))
    for name in node.get("refiners") or []:
        if name not in ob.get("Refiner", {}):
            raise ValueError("Object %s required by Tracker %s was not found" % (name, node["name"]))
        tracker.refiners.append(ob["Refiner"][name])
    tracker.objects = ob
    tracker.bodies = list(ob.get("Body", {}).values())
    tracker.cameras = [c for k in cams for c in ob.get(k, {}).values()]
    tracker.ignored = [n["name"] for k in _IGNORED for n in (d.get(k) or [])]
    return tracker


def _subtree(link):
    out = [link]
    for c in link.children:
        out += _subtree(c)
    return out
