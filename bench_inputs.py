"""Bench / test infrastructure, NOT part of the product package: batches of independent single-body trackers on
seeded synthetic inputs -- what bench.py times, what the parity tests compare, what
__graft_entry__.smoke() runs -- with the worker processes and the on-disk cache that keep their generation out of
the GPU minutes.  `Inputs` holds the rendered frames, sparse viewpoint models and start poses; `Instance` is the object
graph of those inputs behind ONE C-ABI context (the HIP library -- or, in the tests and bench.py's checker legs, the
CPU oracle bound through the same ctypes layer), driven through identical Tracker calls.  (Until round 5 this was
3dobjecttracking_amd/batch.py.)"""
import ctypes as C
import hashlib
import importlib
import os
import pickle
import stat
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_pkg = importlib.import_module("3dobjecttracking_amd")
host = _pkg.host
syn = _pkg.synthetic


def _cache_file(kind, key):
    """M3T_INPUT_CACHE=<dir>: generated inputs are kept there between processes (the models of an 18-model RBOT batch
    take a minute of numpy, its frames another 20 s -- per bench.py invocation, and a profile collection makes dozens).
    The key holds every argument and the generator's own source: a changed generator never meets an old file."""
    root = os.environ.get("M3T_INPUT_CACHE")
    if not root:
        return None
    if not _cache_dir_is_ours(root):
        return None
    h = hashlib.sha1(repr((key, np.__version__, sys.byteorder)).encode())
    for path in (syn.__file__, __file__):
        with open(path, "rb") as f:
            h.update(f.read())
    return os.path.join(root, "%s_%s.pkl" % (kind, h.hexdigest()[:20]))


def _cache_dir_is_ours(root):
    """The cache holds pickles, and loading a pickle runs what is in it: the directory is created 0700 and used only if
    it belongs to this user and nobody else can write to it (a directory somebody else made under /tmp is refused)."""
    try:
        os.makedirs(root, mode=0o700, exist_ok=True)
        st = os.stat(root)
    except OSError:
        return False
    if st.st_uid != os.getuid() or (st.st_mode & (stat.S_IWGRP | stat.S_IWOTH)):
        sys.stderr.write("M3T_INPUT_CACHE=%s is not a private directory of this user: not used\n" % root)
        return False
    return True


def _cache_load(path):
    if path and os.path.exists(path):
        try:
            with open(path, "rb") as f:
                return pickle.load(f)
        except Exception:  # noqa: BLE001 (a torn file from a killed writer: rebuild)
            return None
    return None


def _cache_store(path, obj):
    if not path:
        return
    tmp = "%s.%d.tmp" % (path, os.getpid())
    try:
        with open(tmp, "wb") as f:
            pickle.dump(obj, f, protocol=5)
        os.replace(tmp, path)
    except OSError:  # (disk full, read-only: the cache is a convenience)
        try:
            os.remove(tmp)
        except OSError:
            pass


def _workers(n_tasks, heavy):
    """Processes to generate inputs with.  Opt-in: M3T_INPUT_WORKERS=<n> forces n (1 = in this process),
    M3T_INPUT_WORKERS=auto takes the cores this process may use (affinity, cgroup quota) for inputs that are worth the
    start of a pool; unset = in this process.  (bench.py and the developer tools set `auto` unless told otherwise.)"""
    env = os.environ.get("M3T_INPUT_WORKERS")
    if env is None:
        return 1
    if env != "auto":
        return max(1, min(int(env), n_tasks))
    if not heavy:
        return 1
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    # the ranks of one node generate their inputs at the same time: they share the cores
    try:
        n = max(1, n // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))
    except ValueError:
        pass
    return max(1, min(n, n_tasks))


def _model_task(args):
    body, n_divides, n_points, with_depth = args
    return (syn.make_region_model(body, n_divides=n_divides, n_points=n_points),
            syn.make_depth_model(body, n_divides=n_divides, n_points=n_points) if with_depth else None)


def _scene_task(args):
    """one object's stream: its scene (random state after the last frame included), poses and frames"""
    index, body_index, intr, with_depth, depth_scale, n_frames = args
    sc = syn.Scene(index, intr=intr, with_depth=with_depth, depth_scale=depth_scale)
    if body_index != index:  # an object that shares its model: the other object's shape
        sc.body = syn.Scene(body_index, intr=intr, with_depth=with_depth, depth_scale=depth_scale).body
    gt, color, depth = [], [], []
    for k in range(n_frames):
        if k:
            sc.step_pose()
        gt.append(sc.pose.copy())
        r = sc.render()
        if with_depth:
            color.append(r[0])
            depth.append(r[1])
        else:
            color.append(r)
            depth.append(None)
    return sc, gt, color, depth


def _map(fn, tasks, heavy):
    """fn over tasks, in order; with worker processes where that pays (spawned: the caller may hold a HIP runtime,
    which must not be forked).  Every task is computed by the same code either way: the results are the same bits."""
    import multiprocessing as mp
    import sys
    n = _workers(len(tasks), heavy)
    if mp.current_process().daemon:  # (a worker of somebody else's pool cannot have children)
        n = 1
    if n <= 1 or len(tasks) <= 1:
        return [fn(t) for t in tasks]
    from concurrent.futures import ProcessPoolExecutor
    # the workers need this module only, found by name: keep multiprocessing from running the parent's script again
    # in every worker (spawn does that whenever __main__ has a file or a spec -- a script without a __main__ guard
    # would open its GPU context in each of them)
    main = sys.modules["__main__"]
    hidden = {k: getattr(main, k) for k in ("__file__", "__spec__") if hasattr(main, k)}
    try:
        for k in hidden:
            setattr(main, k, None)
        with ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn")) as pool:
            return list(pool.map(fn, tasks))
    except Exception as e:  # noqa: BLE001 (a worker that died, a box without /dev/shm: the serial path always works)
        sys.stderr.write("input generation: worker processes failed (%s), continuing in this process\n" % (e,))
        return [fn(t) for t in tasks]
    finally:
        for k, v in hidden.items():
            setattr(main, k, v)


class Inputs:
    """Seeded inputs for n_objects independent single-body trackers."""

    def __init__(self, n_objects, n_frames, n_divides=2, n_points=200, intr=None, with_depth=False,
                 depth_scale=1e-4, n_models=None, first_object=0):
        key = (n_objects, n_frames, n_divides, n_points, sorted((intr or {}).items()), with_depth, depth_scale, n_models,
               first_object)
        whole = _cache_file("inputs", key)
        cached = _cache_load(whole)
        if cached is not None:
            self.__dict__.update(cached)
            return
        self._build(n_objects, n_frames, n_divides, n_points, intr, with_depth, depth_scale, n_models, first_object)
        _cache_store(whole, self.__dict__)

    def _build(self, n_objects, n_frames, n_divides, n_points, intr, with_depth, depth_scale, n_models, first_object):
        self.n_objects = n_objects
        self.n_frames = n_frames
        self.with_depth = with_depth
        self.intr = dict(intr or (syn.YCB_INTRINSICS if with_depth else syn.RBOT_INTRINSICS))
        self.depth_scale = depth_scale
        n_models = n_models or n_objects
        self.model_of = [i % n_models for i in range(n_objects)]
        # every object's stream: its scene, ground-truth poses and frames (objects are independent: one task each)
        streams = _map(_scene_task, [(first_object + i, first_object + i % n_models, self.intr, with_depth, depth_scale,
                                      n_frames) for i in range(n_objects)], heavy=n_objects * n_frames >= 256)
        self.scenes = [st[0] for st in streams]
        # objects sharing a model share the body shape
        for i, sc in enumerate(self.scenes):
            if i >= n_models:
                sc.body = self.scenes[i % n_models].body
        # (the models depend on the bodies' shapes only: their own cache entry serves every frame count)
        models_file = _cache_file("models", (first_object, n_models, n_divides, n_points, with_depth,
                                             sorted(self.intr.items()), depth_scale))
        models = _cache_load(models_file)
        if models is None:
            made = _map(_model_task, [(self.scenes[m].body, n_divides, n_points, with_depth) for m in range(n_models)],
                        heavy=n_divides >= 3 and n_models > 1)
            models = ([m[0] for m in made], [m[1] for m in made] if with_depth else None)
            _cache_store(models_file, models)
        self.region_models, self.depth_models = models
        self.gt = [list(st[1]) for st in streams]
        self.color = [list(st[2]) for st in streams]
        self.depth = [list(st[3]) for st in streams]
        # the tracker starts from a slightly wrong pose
        rng = np.random.default_rng(77 + first_object)
        self.start = [syn.perturb_pose(self.gt[i][0], rng, rot_deg=1.0, trans=0.002) for i in range(n_objects)]
        self.vertices = [sc.body.vertices(300, seed=i) for i, sc in enumerate(self.scenes)]


class Instance:
    """The object graph of `inputs` behind one C-ABI context (HIP or oracle)."""

    def __init__(self, api, inputs, region_params=None, depth_params=None, tracker_params=None, use_region=True,
                 use_depth=False, kinds=None):
        # kinds (optional): per object "rd" | "r" | "d" -- which modalities the body carries (default: use_region / use_depth
        # for all of them)
        if kinds is not None:
            use_region, use_depth = any("r" in k for k in kinds), any("d" in k for k in kinds)
        self.api = api
        self.inputs = inputs
        rp = dict(region_params or (syn.YCB_REGION_PARAMS if inputs.with_depth else syn.RBOT_REGION_PARAMS))
        dp = dict(depth_params or syn.YCB_DEPTH_PARAMS)
        tp = dict(tracker_params or (syn.YCB_TRACKER if inputs.with_depth else syn.RBOT_TRACKER))
        if not inputs.with_depth:
            rp["measure_occlusions"] = 0
        intr = inputs.intr
        self.region_models = [host.RegionModel(api, data_points=m[0], orientations=m[1], contour_lengths=m[2])
                              for m in inputs.region_models] if use_region else []
        self.depth_models = [host.DepthModel(api, data_points=m[0], orientations=m[1], surface_areas=m[2])
                             for m in (inputs.depth_models or [])] if use_depth else []
        self.bodies, self.color_cams, self.depth_cams, self.region, self.depth, self.optimizers = [], [], [], [], [], []
        # inputs.camera_of (optional): objects that look at the same frame stream share one camera
        camera_of = getattr(inputs, "camera_of", None)
        shared = {}
        for i in range(inputs.n_objects):
            body = host.Body(api, inputs.start[i])
            if camera_of is not None and camera_of[i] in shared:
                cam, dcam = shared[camera_of[i]]
            else:
                cam = host.ColorCamera(api, **intr)
                dcam = host.DepthCamera(api, depth_scale=inputs.depth_scale, **intr) if inputs.with_depth else None
                if camera_of is not None:
                    shared[camera_of[i]] = (cam, dcam)
            mods = []
            if use_region and (kinds is None or "r" in kinds[i]):
                r = host.RegionModality(api, body, cam, self.region_models[inputs.model_of[i]], depth_camera=dcam, **rp)
                self.region.append(r)
                mods.append(r)
            if use_depth and (kinds is None or "d" in kinds[i]):
                d = host.DepthModality(api, body, dcam, self.depth_models[inputs.model_of[i]], **dp)
                self.depth.append(d)
                mods.append(d)
            self.optimizers.append(host.Optimizer(
                api, body=body, modalities=mods, tikhonov_parameter_rotation=tp["tikhonov_parameter_rotation"],
                tikhonov_parameter_translation=tp["tikhonov_parameter_translation"]))
            self.bodies.append(body)
            self.color_cams.append(cam)
            self.depth_cams.append(dcam)
        self.tracker = host.Tracker(api, tp["n_corr_iterations"], tp["n_update_iterations"])

    def upload_frame(self, k):
        for i in range(self.inputs.n_objects):
            self.color_cams[i].UpdateImage(self.inputs.color[i][k])
            if self.depth_cams[i] is not None:
                self.depth_cams[i].UpdateImage(self.inputs.depth[i][k])

    def poses(self):
        return [b.body2world_pose() for b in self.bodies]

    def set_poses(self, poses):
        for b, p in zip(self.bodies, poses):
            b.set_body2world_pose(p)


def replicate(inputs, n_obj):
    """n_obj objects over the rendered streams of `inputs` (object i looks at stream i mod n_streams through its OWN
    camera and frame ring: distinct device memory, identical content)"""
    if n_obj == inputs.n_objects:
        return inputs
    rep = Inputs.__new__(Inputs)
    rep.__dict__.update(inputs.__dict__)
    idx = [i % inputs.n_objects for i in range(n_obj)]
    rep.n_objects = n_obj
    for name in ("scenes", "model_of", "gt", "color", "depth", "start", "vertices"):
        rep.__dict__[name] = [inputs.__dict__[name][i] for i in idx]
    return rep


def subset(inputs, idx):
    """the objects `idx` of a batch as a batch of their own"""
    sub = Inputs.__new__(Inputs)
    sub.__dict__.update(inputs.__dict__)
    sub.n_objects = len(idx)
    for name in ("scenes", "model_of", "gt", "color", "depth", "start", "vertices"):
        sub.__dict__[name] = [inputs.__dict__[name][i] for i in idx]
    return sub


def stage_frames(api, inst, inputs, n_frames):
    """all frames of all cameras into device-side rings (a new frame is then a pointer switch: cameras_select_slot)"""
    for cams, frames in ((inst.color_cams, inputs.color), (inst.depth_cams, inputs.depth)):
        done = set()
        for i, cam in enumerate(cams):
            if cam is None or cam.id in done:
                continue
            done.add(cam.id)
            api.call("camera_set_ring", cam.id, n_frames)
            for k in range(n_frames):
                f = frames[i][k]
                api.call("camera_upload_slot", cam.id, k, f.ctypes.data_as(C.c_void_p), f.strides[0])
