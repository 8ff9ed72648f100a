"""Synthetic RBOT/YCB-geometry inputs (SURVEY.md §8d "Synthetic inputs").

No dataset is available offline, so benchmark and parity inputs are generated:
bodies are tri-axial ellipsoids (closed-form silhouettes), Sparse Viewpoint
Models hold the exact perspective rim of the ellipsoid seen from the geodesic
view sphere (radius 0.8 m like rbot_evaluator.cpp:548-551), frames are
ray-cast renderings with two noisy colour distributions and, optionally, a u16
depth image.  Seeds follow SURVEY §8d: default_rng(1000 + object_index).

This is input generation only (numpy on the host); nothing here is part of the
measured path.
"""
import numpy as np

from ._capi import M3T_DEPTH_POINT_FLOATS, M3T_REGION_POINT_FLOATS

FLT_MAX = np.float32(3.4028234663852886e38)

# RBOT camera (M3T/examples/rbot_evaluator.h:40-41)
RBOT_INTRINSICS = dict(fu=650.048, fv=647.183, ppu=323.828, ppv=256.823, width=640, height=512)
# YCB-Video camera (M3T/examples/ycb_evaluator.h:47-48)
YCB_INTRINSICS = dict(fu=1066.778, fv=1067.487, ppu=312.9869, ppv=241.3109, width=640, height=480)

# RBOT parameter set (M3T/examples/evaluate_rbot_dataset.cpp:25-44,76-83)
RBOT_REGION_PARAMS = dict(
    n_lines_max=200, min_continuous_distance=3.0, function_length=8, distribution_length=12,
    function_amplitude=0.36, function_slope=0.0, learning_rate=1.3, n_global_iterations=1,
    scales=[5, 2, 2, 1], standard_deviations=[20.0, 7.0, 3.0, 1.5], n_histogram_bins=32,
    learning_rate_f=0.2, learning_rate_b=0.2, unconsidered_line_length=0.5,
    max_considered_line_length=20.0, n_unoccluded_iterations=0)
RBOT_TRACKER = dict(n_corr_iterations=7, n_update_iterations=2, tikhonov_parameter_rotation=1000.0,
                    tikhonov_parameter_translation=30000.0)
# YCB parameter set (M3T/examples/evaluate_ycb_dataset.cpp:46-76,108-115)
YCB_REGION_PARAMS = dict(
    n_lines_max=200, min_continuous_distance=3.0, function_length=8, distribution_length=12,
    function_amplitude=0.43, function_slope=0.5, learning_rate=1.3, n_global_iterations=1,
    scales=[7, 4, 2], standard_deviations=[25.0, 15.0, 10.0], n_histogram_bins=16,
    learning_rate_f=0.2, learning_rate_b=0.2, unconsidered_line_length=0.5,
    max_considered_line_length=20.0, measure_occlusions=1, measured_depth_offset_radius=0.01,
    measured_occlusion_radius=0.01, measured_occlusion_threshold=0.03, n_unoccluded_iterations=0,
    min_n_unoccluded_lines=0)
YCB_DEPTH_PARAMS = dict(
    n_points_max=200, stride_length=0.005, considered_distances=[0.07, 0.05, 0.04],
    standard_deviations=[0.05, 0.03, 0.02], measure_occlusions=1, measured_depth_offset_radius=0.01,
    measured_occlusion_radius=0.01, measured_occlusion_threshold=0.03, n_unoccluded_iterations=0,
    min_n_unoccluded_points=0)
YCB_TRACKER = dict(n_corr_iterations=4, n_update_iterations=2, tikhonov_parameter_rotation=1000.0,
                   tikhonov_parameter_translation=30000.0)


def geodesic_points(n_divides):
    """Unit vectors of an icosphere subdivided n_divides times: 10*4^n + 2 points
    (what Model::GenerateGeodesicPoints produces, model.cpp:412-454; order is ours)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
             (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(n_divides):
        cache = {}

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.asarray(verts, np.float64)


def _basis(d):
    """two unit vectors orthogonal to each row of d (N,3)"""
    a = np.where(np.abs(d[:, :1]) < 0.9, np.array([[1.0, 0, 0]]), np.array([[0, 1.0, 0]]))
    e1 = np.cross(d, a)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.cross(d, e1)
    return e1, e2


class Ellipsoid:
    """x^T diag(1/a^2,1/b^2,1/c^2) x = 1 in the body frame."""

    def __init__(self, semi_axes):
        self.s = np.asarray(semi_axes, np.float64)

    def vertices(self, n=400, seed=0):
        rng = np.random.default_rng(seed)
        y = rng.normal(size=(n, 3))
        y /= np.linalg.norm(y, axis=1, keepdims=True)
        return (y * self.s).astype(np.float32)

    def rim(self, cam_center, phi):
        """Exact perspective contour generator seen from cam_center (N,3) (body frame):
        returns points (N,P,3) and outward unit surface normals (N,P,3)."""
        cp = cam_center / self.s  # sphere space
        n2 = np.sum(cp * cp, axis=1, keepdims=True)
        centre = cp / n2
        r = np.sqrt(np.maximum(1.0 - 1.0 / n2, 0.0))
        e1, e2 = _basis(cp / np.sqrt(n2))
        y = (centre[:, None, :] + r[:, None, :] * (np.cos(phi)[..., None] * e1[:, None, :] +
                                                     np.sin(phi)[..., None] * e2[:, None, :]))
        x = y * self.s
        nrm = y / self.s
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        return x, nrm

    def visible_points(self, cam_center, rng, n_points):
        """random surface points visible from cam_center (N,3): (N,P,3) points + normals"""
        cp = cam_center / self.s
        n = np.linalg.norm(cp, axis=1, keepdims=True)
        d = cp / n
        e1, e2 = _basis(d)
        N = cam_center.shape[0]
        # cap: d.y in (1/n, 1]; keep a margin so normals are not grazing
        lo = (1.0 / n) + 0.15 * (1.0 - 1.0 / n)
        h = lo + (1.0 - lo) * rng.random((N, n_points))
        ang = 2 * np.pi * rng.random((N, n_points))
        rr = np.sqrt(np.maximum(1.0 - h * h, 0.0))
        y = (h[..., None] * d[:, None, :] + rr[..., None] * (np.cos(ang)[..., None] * e1[:, None, :] +
                                                            np.sin(ang)[..., None] * e2[:, None, :]))
        x = y * self.s
        nrm = y / self.s
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        return x, nrm


def _ray_hits(part_s, o, d):
    """does the ray o + s d (s > 0) hit the ellipsoid with semi axes part_s (arrays (...,3))"""
    ob, db = o / part_s, d / part_s
    a = np.sum(db * db, -1)
    b = 2 * np.sum(db * ob, -1)
    c = np.sum(ob * ob, -1) - 1.0
    disc = b * b - 4 * a * c
    s1 = (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a)
    return (disc > 0) & (s1 > 0), s1


class Composite:
    """Union of ellipsoids (each with its own pose in the body frame): a non-convex, non-symmetric
    body whose silhouette constrains all six degrees of freedom (a single ellipsoid does not)."""

    def __init__(self, parts):
        self.parts = [(Ellipsoid(s), np.asarray(R, np.float64), np.asarray(t, np.float64)) for s, R, t in parts]
        self.s = np.array([max(np.linalg.norm(t) + np.max(e.s) for e, R, t in self.parts)] * 3)  # bounding radius

    def vertices(self, n=400, seed=0):
        out = []
        for k, (e, R, t) in enumerate(self.parts):
            v = e.vertices(n // len(self.parts), seed + k).astype(np.float64)
            out.append(v @ R.T + t)
        return np.concatenate(out).astype(np.float32)

    def hits(self, o, d, skip=None):
        """ray o + s d against all parts (body frame, arrays (...,3)): (hit mask, nearest s)"""
        hit = np.zeros(o.shape[:-1], bool)
        smin = np.full(o.shape[:-1], np.inf)
        for k, (e, R, t) in enumerate(self.parts):
            if k == skip:
                continue
            h, s1 = _ray_hits(e.s, (o - t) @ R, d @ R)
            smin = np.where(h & (s1 < smin), s1, smin)
            hit |= h
        return hit, smin

    def rim_candidates(self, cam, phi):
        """per part exact rims seen from cam (V,3): points, normals (V, K*P, 3), own chord (V, K*P), valid mask"""
        xs, ns, chords, valid = [], [], [], []
        for k, (e, R, t) in enumerate(self.parts):
            cp = (cam - t) @ R
            x, n = e.rim(cp, phi)
            chords.append(_ellipse_chord(e, cp, x, n))
            xb = x @ R.T + t
            nb = n @ R.T
            d = xb - cam[:, None, :]
            h, _ = self.hits(np.broadcast_to(cam[:, None, :], xb.shape), d, skip=k)
            xs.append(xb)
            ns.append(nb)
            valid.append(~h)
        return np.concatenate(xs, 1), np.concatenate(ns, 1), np.concatenate(chords, 1), np.concatenate(valid, 1)


def _ellipse_chord(e, cam_p, x, nrm):
    """length of the chord of the rim ellipse of `e` (seen from cam_p, part frame) from the rim
    point x along the inward normal, in metres (orthographic footprint of the rim)"""
    V = cam_p.shape[0]
    ori = -cam_p / np.linalg.norm(cam_p, axis=1, keepdims=True)
    e1, e2 = _basis(ori)
    xc, _ = e.rim(cam_p, np.zeros((V, 1)))
    xs, _ = e.rim(cam_p, np.full((V, 1), np.pi / 2))
    xo, _ = e.rim(cam_p, np.full((V, 1), np.pi))
    centre3 = 0.5 * (xc[:, 0] + xo[:, 0])
    m1 = xc[:, 0] - centre3
    m2 = xs[:, 0] - centre3
    M = np.stack([np.stack([np.sum(m1 * e1, 1), np.sum(m2 * e1, 1)], 1),
                  np.stack([np.sum(m1 * e2, 1), np.sum(m2 * e2, 1)], 1)], 1)
    Minv = np.linalg.inv(M)
    p0 = np.stack([np.sum((x - centre3[:, None]) * e1[:, None], 2), np.sum((x - centre3[:, None]) * e2[:, None], 2)], 2)
    n2 = np.stack([np.sum(nrm * e1[:, None], 2), np.sum(nrm * e2[:, None], 2)], 2)
    n2 /= np.linalg.norm(n2, axis=2, keepdims=True)
    q0 = np.einsum("vij,vpj->vpi", Minv, p0)
    w = np.einsum("vij,vpj->vpi", Minv, n2)
    return np.maximum(2.0 * np.sum(q0 * w, 2) / np.sum(w * w, 2), 0.0)


def make_composite_region_model(body, n_divides=4, n_points=200, sphere_radius=0.8, seed=7):
    """Sparse viewpoint model of a Composite: rim points of every part that lie on the union's
    silhouette, sampled at random (like the reference's mt19937 sampling of the contour)."""
    rng = np.random.default_rng(seed)
    ori = geodesic_points(n_divides)
    V = ori.shape[0]
    cam = -sphere_radius * ori
    P = 2 * n_points
    phi = (np.arange(P)[None, :] + rng.random((V, 1))) * (2 * np.pi / P)
    x, nrm, chord, valid = body.rim_candidates(cam, phi)
    # random choice of n_points valid candidates per view
    score = rng.random(valid.shape) + (~valid) * 10.0
    pick = np.argsort(score, axis=1)[:, :n_points]
    n_valid = valid.sum(1)
    assert n_valid.min() >= 8, "degenerate view"
    # views with fewer valid candidates than n_points: wrap around the valid ones
    wrap = np.arange(n_points)[None, :] % np.maximum(n_valid, 1)[:, None]
    pick = np.take_along_axis(pick, wrap, axis=1)
    take = lambda a: np.take_along_axis(a, pick[..., None] if a.ndim == 3 else pick, axis=1)
    x, nrm, chord = take(x), take(nrm), take(chord)
    # background distance: first re-entry into the body along the outward normal (view ray through x + t n)
    bg = np.full(x.shape[:2], np.float64(FLT_MAX))
    camb = np.broadcast_to(cam[:, None, :], x.shape)
    for tstep in np.arange(0.006, 0.10, 0.008)[::-1]:
        h, _ = body.hits(camb, x + tstep * nrm - camb)
        bg = np.where(h, tstep, bg)
    take = lambda a: a
    dp = np.zeros((V, n_points, M3T_REGION_POINT_FLOATS), np.float32)
    dp[:, :, 0:3] = take(x)
    dp[:, :, 3:6] = take(nrm)
    dp[:, :, 6] = take(chord)
    dp[:, :, 7] = np.minimum(take(bg), FLT_MAX).astype(np.float32)
    contour = (2 * np.pi * np.mean(body.s) * n_valid / valid.shape[1]).astype(np.float32)
    return dp, ori.astype(np.float32), contour


def make_composite_depth_model(body, n_divides=4, n_points=200, sphere_radius=0.8, seed=7):
    rng = np.random.default_rng(seed + 1)
    ori = geodesic_points(n_divides)
    V = ori.shape[0]
    cam = -sphere_radius * ori
    xs, ns, valid = [], [], []
    for k, (e, R, t) in enumerate(body.parts):
        x, n = e.visible_points((cam - t) @ R, rng, 2 * n_points)
        xb, nb = x @ R.T + t, n @ R.T
        camb = np.broadcast_to(cam[:, None, :], xb.shape)
        h, s1 = body.hits(camb, xb - camb, skip=k)
        xs.append(xb)
        ns.append(nb)
        valid.append(~(h & (s1 < 1.0)))
    x, nrm, valid = np.concatenate(xs, 1), np.concatenate(ns, 1), np.concatenate(valid, 1)
    score = rng.random(valid.shape) + (~valid) * 10.0
    pick = np.argsort(score, axis=1)[:, :n_points]
    dp = np.zeros((V, n_points, M3T_DEPTH_POINT_FLOATS), np.float32)
    dp[:, :, 0:3] = np.take_along_axis(x, pick[..., None], axis=1)
    dp[:, :, 3:6] = np.take_along_axis(nrm, pick[..., None], axis=1)
    return dp, ori.astype(np.float32), np.full(V, np.pi * np.mean(body.s) ** 2, np.float32)


def make_region_model(body, n_divides=4, n_points=200, sphere_radius=0.8, seed=7):
    """Sparse viewpoint model arrays: data_points (V,P,38), orientations (V,3), contour_lengths (V)."""
    if isinstance(body, Composite):
        return make_composite_region_model(body, n_divides, n_points, sphere_radius, seed)
    rng = np.random.default_rng(seed)
    ori = geodesic_points(n_divides)  # camera -> body direction, body frame
    V = ori.shape[0]
    cam = -sphere_radius * ori
    phi = (np.arange(n_points)[None, :] + rng.random((V, 1))) * (2 * np.pi / n_points)
    # the reference samples contour points in random order (mt19937 % N): shuffle
    perm = np.argsort(rng.random((V, n_points)), axis=1)
    phi = np.take_along_axis(phi, perm, axis=1)
    x, nrm = body.rim(cam, phi)
    # image-plane basis of each view (camera looks along ori)
    e1, e2 = _basis(ori)
    # orthographic footprint of the rim ellipse: p = C + M [cos, sin]
    xc, _ = body.rim(cam, np.zeros((V, 1)))
    xs, _ = body.rim(cam, np.full((V, 1), np.pi / 2))
    xo, _ = body.rim(cam, np.full((V, 1), np.pi))
    centre3 = 0.5 * (xc[:, 0] + xo[:, 0])
    m1 = xc[:, 0] - centre3
    m2 = xs[:, 0] - centre3
    M = np.stack([np.stack([np.sum(m1 * e1, 1), np.sum(m2 * e1, 1)], 1),
                  np.stack([np.sum(m1 * e2, 1), np.sum(m2 * e2, 1)], 1)], 1)  # (V,2,2)
    Minv = np.linalg.inv(M)
    p0 = np.stack([np.sum((x - centre3[:, None]) * e1[:, None], 2), np.sum((x - centre3[:, None]) * e2[:, None], 2)], 2)
    n2 = np.stack([np.sum(nrm * e1[:, None], 2), np.sum(nrm * e2[:, None], 2)], 2)
    n2 /= np.linalg.norm(n2, axis=2, keepdims=True)
    q0 = np.einsum("vij,vpj->vpi", Minv, p0)
    w = np.einsum("vij,vpj->vpi", Minv, n2)
    chord = 2.0 * np.sum(q0 * w, 2) / np.sum(w * w, 2)
    dp = np.zeros((V, n_points, M3T_REGION_POINT_FLOATS), np.float32)
    dp[:, :, 0:3] = x
    dp[:, :, 3:6] = nrm
    dp[:, :, 6] = np.maximum(chord, 0.0)
    dp[:, :, 7] = FLT_MAX
    # perimeter (Ramanujan) of the footprint ellipse
    sv = np.linalg.svd(M, compute_uv=False)
    a, b = sv[:, 0], sv[:, 1]
    contour = np.pi * (3 * (a + b) - np.sqrt((3 * a + b) * (a + 3 * b)))
    return dp, ori.astype(np.float32), contour.astype(np.float32)


def make_depth_model(body, n_divides=4, n_points=200, sphere_radius=0.8, seed=7):
    if isinstance(body, Composite):
        return make_composite_depth_model(body, n_divides, n_points, sphere_radius, seed)
    rng = np.random.default_rng(seed + 1)
    ori = geodesic_points(n_divides)
    V = ori.shape[0]
    cam = -sphere_radius * ori
    x, nrm = body.visible_points(cam, rng, n_points)
    dp = np.zeros((V, n_points, M3T_DEPTH_POINT_FLOATS), np.float32)
    dp[:, :, 0:3] = x
    dp[:, :, 3:6] = nrm
    area = np.full(V, np.pi * np.prod(body.s) ** (2.0 / 3.0), np.float32)
    return dp, ori.astype(np.float32), area


def rot_vec(r):
    """Rodrigues: rotation vector -> 3x3 (float64)"""
    r = np.asarray(r, np.float64)
    th = np.linalg.norm(r)
    if th < 1e-12:
        return np.eye(3)
    k = r / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def make_pose(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def random_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


class Scene:
    """One object with its own camera stream: background texture, colour means and GT trajectory."""

    def __init__(self, object_index, intr=RBOT_INTRINSICS, semi_axes=None, with_depth=False, depth_scale=1e-4,
                 distance=(0.5, 0.8)):
        self.rng = np.random.default_rng(1000 + object_index)
        rng = self.rng
        self.intr = dict(intr)
        if semi_axes is None:
            # two ellipsoidal lobes at an angle: all six degrees of freedom show in the silhouette
            a = rng.uniform(0.06, 0.085)
            main = a * np.array([1.0, rng.uniform(0.4, 0.55), rng.uniform(0.28, 0.4)])
            lobe = a * np.array([rng.uniform(0.6, 0.8), rng.uniform(0.28, 0.4), rng.uniform(0.22, 0.32)])
            R_l = rot_vec(rng.uniform(0.6, 1.2) * np.array([0.3, 0.4, 1.0]) * rng.choice([-1, 1]))
            t_l = a * np.array([rng.uniform(0.45, 0.7), rng.uniform(0.25, 0.45) * rng.choice([-1, 1]), rng.uniform(-0.15, 0.15)])
            self.body = Composite([(main, np.eye(3), np.zeros(3)), (lobe, R_l, t_l)])
        else:
            self.body = Ellipsoid(semi_axes)
        W, H = intr["width"], intr["height"]
        mu_b = rng.uniform(40, 215, 3)
        mu_f = (mu_b + rng.choice([-1, 1], 3) * rng.uniform(60, 110, 3)) % 256
        self.mu_f, self.mu_b = mu_f, mu_b
        coarse = rng.normal(0, 1, (H // 8 + 2, W // 8 + 2, 3))
        tex = np.kron(coarse, np.ones((8, 8, 1)))[:H, :W]
        self.background = np.clip(mu_b + 18.0 * tex + rng.normal(0, 12, (H, W, 3)), 0, 255).astype(np.uint8)
        z = rng.uniform(*distance)
        u = rng.uniform(0.3 * W, 0.7 * W)
        v = rng.uniform(0.3 * H, 0.7 * H)
        t = np.array([(u - intr["ppu"]) * z / intr["fu"], (v - intr["ppv"]) * z / intr["fv"], z])
        self.pose = make_pose(random_rotation(rng), t)  # body2camera == body2world (camera at origin)
        self.with_depth = with_depth
        self.depth_scale = depth_scale
        self.background_depth = rng.uniform(1.0, 1.6)

    def step_pose(self, max_rot_deg=1.0, max_trans=0.003):
        """GT motion between frames: U(+-1 deg) rotation per axis, U(+-3 mm) translation
        (SURVEY §8d proposes +-2 deg; the ellipsoid silhouettes constrain two rotations only weakly)."""
        r = self.rng.uniform(-1, 1, 3) * np.deg2rad(max_rot_deg)
        dt = self.rng.uniform(-1, 1, 3) * max_trans
        self.pose = make_pose(self.pose[:3, :3] @ rot_vec(r), self.pose[:3, 3] + dt)
        return self.pose

    def render(self, pose=None):
        """-> BGR8 (H,W,3) [, u16 depth (H,W)] of the ellipsoid at `pose` (body2camera)."""
        pose = self.pose if pose is None else pose
        intr = self.intr
        W, H = intr["width"], intr["height"]
        img = self.background.copy()
        R, t = pose[:3, :3], pose[:3, 3]
        rmax = float(np.max(self.body.s))
        zc = t[2]
        uc = t[0] * intr["fu"] / zc + intr["ppu"]
        vc = t[1] * intr["fv"] / zc + intr["ppv"]
        rad = rmax * intr["fu"] / max(zc - rmax, 1e-3) + 3
        u0, u1 = int(max(0, np.floor(uc - rad))), int(min(W, np.ceil(uc + rad) + 1))
        v0, v1 = int(max(0, np.floor(vc - rad))), int(min(H, np.ceil(vc + rad) + 1))
        depth = None
        if self.with_depth:
            depth = np.full((H, W), self.background_depth, np.float64)
        if u1 > u0 and v1 > v0:
            uu, vv = np.meshgrid(np.arange(u0, u1), np.arange(v0, v1))
            d = np.stack([(uu - intr["ppu"]) / intr["fu"], (vv - intr["ppv"]) / intr["fv"], np.ones_like(uu, float)], -1)
            # ray o + s d in the body frame (d has z = 1 in the camera frame -> s is the depth)
            db = d @ R
            ob = np.broadcast_to(-(R.T @ t), db.shape)
            if isinstance(self.body, Composite):
                hit, s_hit = self.body.hits(ob, db)
            else:
                hit, s_hit = _ray_hits(self.body.s, ob, db)
            nh = int(hit.sum())
            if nh:
                sub = img[v0:v1, u0:u1]
                sub[hit] = np.clip(self.mu_f + self.rng.normal(0, 20, (nh, 3)), 0, 255).astype(np.uint8)
                if depth is not None:
                    dsub = depth[v0:v1, u0:u1]
                    dsub[hit] = s_hit[hit]
        if depth is not None:
            noisy = depth + self.rng.normal(0, 0.001, depth.shape)
            d16 = np.clip(noisy / self.depth_scale, 0, 65535).astype(np.uint16)
            return img, d16
        return img


def perturb_pose(pose, rng, rot_deg=2.0, trans=0.003):
    r = rng.uniform(-1, 1, 3) * np.deg2rad(rot_deg)
    dt = rng.uniform(-1, 1, 3) * trans
    return make_pose(pose[:3, :3] @ rot_vec(r), pose[:3, 3] + dt)


def pose_errors(a, b):
    """rotation geodesic error (rad) and translation error (m) between two 4x4 poses"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    Rd = a[:3, :3].T @ b[:3, :3]
    # atan2(sin, cos): arccos alone has a ~3e-4 rad noise floor for float32 matrices
    sn = 0.5 * np.linalg.norm([Rd[2, 1] - Rd[1, 2], Rd[0, 2] - Rd[2, 0], Rd[1, 0] - Rd[0, 1]])
    cs = (np.trace(Rd) - 1) / 2
    return float(np.arctan2(sn, cs)), float(np.linalg.norm(a[:3, 3] - b[:3, 3]))


def add_s(vertices, a, b):
    """ADD-S as ycb_evaluator.cpp:816-831: mean nearest-neighbour distance between the
    model vertices under pose a and under pose b (brute force instead of nanoflann)."""
    v = np.asarray(vertices, np.float64)
    pa = v @ np.asarray(a, np.float64)[:3, :3].T + np.asarray(a, np.float64)[:3, 3]
    pb = v @ np.asarray(b, np.float64)[:3, :3].T + np.asarray(b, np.float64)[:3, 3]
    d2 = np.sum((pa[:, None, :] - pb[None, :, :]) ** 2, -1)
    return float(np.mean(np.sqrt(np.min(d2, axis=1))))
