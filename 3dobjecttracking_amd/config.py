"""Readers for the reference's on-disk configuration: OpenCV-FileStorage YAML metafiles, Wavefront meshes
and the sparse-viewpoint-model .bin header.

The reference reads every metafile through cv::FileStorage (common.cpp:84-100): a '%YAML:1.2' directive
line, plain scalars / quoted strings, flow sequences and '!!opencv-matrix' nodes (rows, cols, dt, row-major
data) for the 4x4 poses.  Keys are the member names without the trailing underscore; paths inside a metafile
are relative to the metafile's directory (body.cpp:176-178, loader_camera.cpp:141-144, model.cpp:233-234).
"""
import os
import struct

import numpy as np
import yaml


class _Loader(yaml.SafeLoader):
    pass


def _opencv_matrix(loader, node):
    m = loader.construct_mapping(node, deep=True)
    return np.asarray(m["data"], np.float64).reshape(int(m["rows"]), int(m["cols"]))


_Loader.add_constructor("tag:yaml.org,2002:opencv-matrix", _opencv_matrix)


def read_yaml(path):
    """OpenYamlFileStorage (common.cpp:84-100): dict of the top-level nodes; raises FileNotFoundError /
    ValueError where the reference prints and returns false."""
    with open(path) as f:
        text = f.read()
    if text.startswith("%YAML"):  # '%YAML:1.2' is OpenCV's spelling of the directive, not YAML's
        text = text.split("\n", 1)[1] if "\n" in text else ""
    try:
        d = yaml.load(text, Loader=_Loader)
    except yaml.YAMLError as e:
        raise ValueError("Could not open file %s: %s" % (path, e))
    return d if d is not None else {}


def required(d, keys, what, path):
    missing = [k for k in keys if k not in d]
    if missing:
        raise ValueError("Could not read all required %s parameters from %s (missing %s)" % (what, path, missing))


def relative_to(metafile_path, p):
    return p if os.path.isabs(p) else os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(metafile_path)), p))


def pose(value):
    """Transform3fA of a yaml node (row-major 4x4, read as double, stored as float: common.h:231-240)"""
    m = np.asarray(value, np.float64).reshape(4, 4)
    return m.astype(np.float32)


def load_obj(path, geometry_unit_in_meter=1.0):
    """Body::LoadMeshData (body.cpp:185-242) for the records tiny_obj_loader hands over: 'v x y z' vertices
    (scaled to metres) and 'f' faces by vertex index (1-based, negative = relative to the end).  Polygons with
    more than three corners are split as a fan, which is what tiny_obj_loader's triangulation does for convex
    faces.  Returns (vertices [n,3] f32, triangles [m,3] i32) in the file's winding."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t or t[0].startswith("#"):
                continue
            if t[0] == "v":
                verts.append((float(t[1]), float(t[2]), float(t[3])))
            elif t[0] == "f":
                idx = []
                for corner in t[1:]:
                    i = int(corner.split("/")[0])
                    idx.append(i - 1 if i > 0 else len(verts) + i)
                for k in range(1, len(idx) - 1):
                    faces.append((idx[0], idx[k], idx[k + 1]))
    if not verts or not faces:
        raise ValueError("TinyObjLoader failed to load data from %s" % path)
    v = np.asarray(verts, np.float32)
    if geometry_unit_in_meter != 1.0:
        v = v * np.float32(geometry_unit_in_meter)
    return v, np.asarray(faces, np.int32)


def maximum_body_diameter(vertices):
    """Body::CalculateMaximumBodyDiameter (body.cpp:244-252): twice the largest vertex norm"""
    v = np.asarray(vertices, np.float32)
    return float(np.float32(2.0) * np.sqrt((v * v).sum(axis=1, dtype=np.float32)).max())


# ---------------------------------------------------------------------------------------------------------
# sparse viewpoint model files (model.cpp:218-323, region_model.cpp:259-363, depth_model.cpp:215-291)
# ---------------------------------------------------------------------------------------------------------
REGION_VERSION, DEPTH_VERSION = 10, 9


class BodyData:
    """what a model file records about a body (model.cpp:301-323)"""

    def __init__(self, geometry_path, geometry_unit_in_meter, geometry_counterclockwise, geometry_enable_culling,
                 maximum_body_diameter, geometry2body_pose):
        self.geometry_path = str(geometry_path)
        self.geometry_unit_in_meter = np.float32(geometry_unit_in_meter)
        self.geometry_counterclockwise = bool(geometry_counterclockwise)
        self.geometry_enable_culling = bool(geometry_enable_culling)
        self.maximum_body_diameter = np.float32(maximum_body_diameter)
        self.geometry2body_pose = np.asarray(geometry2body_pose, np.float32).reshape(4, 4)

    def pack(self):
        p = self.geometry_path.encode()
        return (struct.pack("<Q", len(p)) + p +
                struct.pack("<f??f", self.geometry_unit_in_meter, self.geometry_counterclockwise,
                            self.geometry_enable_culling, self.maximum_body_diameter) +
                self.geometry2body_pose.T.astype("<f4").tobytes())  # Eigen storage: column-major

    @classmethod
    def unpack(cls, b, off):
        (n,) = struct.unpack_from("<Q", b, off)
        off += 8
        path = b[off:off + n].decode()
        off += n
        unit, ccw, cull, diam = struct.unpack_from("<f??f", b, off)
        off += 10
        g2b = np.frombuffer(b, "<f4", 16, off).reshape(4, 4).T.copy()
        return cls(path, unit, ccw, cull, diam, g2b), off + 64

    def __eq__(self, o):
        return (_equivalent(self.geometry_path, o.geometry_path) and
                self.geometry_unit_in_meter == o.geometry_unit_in_meter and
                self.geometry_counterclockwise == o.geometry_counterclockwise and
                self.geometry_enable_culling == o.geometry_enable_culling and
                self.maximum_body_diameter == o.maximum_body_diameter and
                np.array_equal(self.geometry2body_pose, o.geometry2body_pose))


def _equivalent(a, b):
    """common.cpp Equivalent(): the same file, or (if one does not exist) the same normalised path"""
    try:
        return os.path.samefile(a, b)
    except OSError:
        return os.path.normpath(a) == os.path.normpath(b)


def _model_parameters(p):
    return struct.pack("<fiiff?i", p["sphere_radius"], p["n_divides"], p["n_points"], p["max_radius_depth_offset"],
                       p["stride_depth_offset"], bool(p["use_random_seed"]), p["image_size"])


def write_model_bin(path, region, params, body_data, points, orientations, extents, associated=None):
    """RegionModel / DepthModel::SaveModel: header, body data, associated-body lists (empty unless given),
    then per view the data points, the orientation and the contour length / surface area."""
    points = np.ascontiguousarray(points, "<f4")
    n_views, n_points = points.shape[0], points.shape[1]
    assert n_points == params["n_points"]
    head = struct.pack("<ci", b"r" if region else b"d", REGION_VERSION if region else DEPTH_VERSION)
    head += _model_parameters(params) + body_data.pack()
    groups = list(associated) if associated is not None else ([[], [], [], []] if region else [[]])
    if region:
        head += struct.pack("<Q", sum(len(g) for g in groups))
    for g in groups:
        head += struct.pack("<Q", len(g)) + b"".join(bd.pack() for bd in g)
    head += struct.pack("<Q", n_views)
    rec = np.zeros((n_views, points.shape[2] * n_points + 4), "<f4")
    rec[:, :-4] = points.reshape(n_views, -1)
    rec[:, -4:-1] = np.asarray(orientations, "<f4").reshape(n_views, 3)
    rec[:, -1] = np.asarray(extents, "<f4").reshape(n_views)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(head)
        f.write(rec.tobytes())


def model_bin_matches(path, region, params, body_data, associated=None):
    """the acceptance test of Model::LoadModelParameters / LoadBodyData (model.cpp:237-284): same type and
    version, same generation parameters (a file with more points per view is refused here: its views could
    not be read with the requested stride), same main body and associated bodies"""
    try:
        with open(path, "rb") as f:
            b = f.read(1 << 20)
    except OSError:
        return False
    try:
        kind, version = struct.unpack_from("<ci", b, 0)
        if kind != (b"r" if region else b"d") or version != (REGION_VERSION if region else DEPTH_VERSION):
            return False
        if b[5:30] != _model_parameters(params):
            return False
        bd, off = BodyData.unpack(b, 30)
        if not bd == body_data:
            return False
        groups = list(associated) if associated is not None else ([[], [], [], []] if region else [[]])
        if region:
            (n,) = struct.unpack_from("<Q", b, off)
            off += 8
            if n != sum(len(g) for g in groups):
                return False
        for g in groups:
            (n,) = struct.unpack_from("<Q", b, off)
            off += 8
            if n != len(g):
                return False
            for want in g:
                have, off = BodyData.unpack(b, off)
                if not have == want:
                    return False
        return True
    except (struct.error, ValueError, UnicodeDecodeError):
        return False
