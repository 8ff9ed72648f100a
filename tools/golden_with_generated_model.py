"""Developer check: the reference's region goldens evaluated by the oracle with the triangle model the
product generated (m3t_hip_region_model_generate) instead of the numpy prototype's."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: F401

torch.cuda.init()
import gl_model as g  # noqa: E402
import golden_scene as gs  # noqa: E402
import util  # noqa: E402
from util import host  # noqa: E402

hip = util.open_hip()
tv, tf = g.load_obj(os.path.join(util.GOLDEN, "_body/triangle.obj"))
body = host.Body(hip, gs.mtv.body2world())
body.set_geometry(tv, tf, np.asarray(gs.mtv.GEOMETRY2BODY, np.float32), body_id=150, region_id=150)
region = host.RegionModel.generate(hip, body)
rp, ro, rl = region.views()
fx = gs.views()
views = fx["region_views"]
np.savez(os.path.join(ROOT, "gpurun_out", "generated_triangle_region_views.npz"), points=rp[views], orientations=ro[views],
         contour_lengths=rl[views])
api = util.open_oracle()
for name, (P, O, L) in (("prototype", (fx["region_points"], fx["region_orientations"], fx["region_contour_lengths"])),
                        ("product", (rp[views], ro[views], rl[views]))):
    api = util.open_oracle()
    model = host.RegionModel(api, data_points=P, orientations=O, contour_lengths=L)
    b = host.Body(api, gs.mtv.body2world())
    cam = host.ColorCamera(api, **gs.mtv.COLOR_INTRINSICS)
    mod = host.RegionModality(api, b, cam, model)
    host.Optimizer(api, body=b, modalities=[mod])
    img = gs.load_png("_sequence/color_camera_image_200.png")
    cam.UpdateImage(img)
    t = host.Tracker(api)
    t.StartModalities(0)
    t.CalculateCorrespondences(0, 0)
    lines = mod.data_lines()
    hf, hb = mod.histograms()
    vis = gs.render_lines_visualisation(img, hf, hb, lines, 16, 6, 12)
    gold = gs.load_png("modality_test/region_modality.png").astype(np.int32)
    out = [int((np.abs(vis - gold).max(axis=2) > 0).sum())]
    for it, nm in ((0, "global"), (1, "local")):
        t.CalculateGradientAndHessian(0, 0, it)
        gr, h = mod.gradient_hessian()
        gg = gs.golden(f"region_modality_{nm}_gradient.txt")[:, 0]
        out.append(float(np.linalg.norm(gr - gg) / np.linalg.norm(gg)))
        out.append(float(np.max(np.abs((gr - gg) / gg))))
    print(name, "vis px diff %d | global |dg|/|g| %.2e max rel %.2e | local %.2e max rel %.2e" % tuple(out))
