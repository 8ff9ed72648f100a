"""The focused renderers and the renderer-fed branches on the device against the oracle
(bit-exact: the rasterisation is integer arithmetic on snapped coordinates plus f64 depth
interpolation; atomicMin on packed words makes the z-buffer independent of the triangle order)."""
import numpy as np
import pytest

import golden_scene as gs
import util
from util import host

pytestmark = pytest.mark.gpu


def _scene(api, image_size):
    f = gs.TrackerFixture(api, measure_occlusions=False, region_params=dict(n_unoccluded_iterations=0),
                          depth_params=dict(n_unoccluded_iterations=0))
    geometry, schauma = gs.fixture_renderer_geometry(api, f.body)
    r = dict(
        color_depth=host.FocusedBasicDepthRenderer(api, geometry, f.color_camera, image_size=image_size),
        color_sil=host.FocusedSilhouetteRenderer(api, geometry, f.color_camera, id_type=1, image_size=image_size),
        depth_depth=host.FocusedBasicDepthRenderer(api, geometry, f.depth_camera, image_size=image_size),
        depth_sil=host.FocusedSilhouetteRenderer(api, geometry, f.depth_camera, id_type=0, image_size=image_size))
    for x in r.values():
        x.AddReferencedBody(f.body)
    return f, schauma, r


@pytest.mark.parametrize("image_size", [200, 64, 333])  # 333: z-buffer in HBM instead of LDS
def test_renderings_match_oracle(image_size):
    out = []
    for api in (util.open_hip(), util.open_oracle()):
        f, schauma, r = _scene(api, image_size)
        images = []
        for name in sorted(r):
            r[name].StartRendering()
            images.append(r[name].images())
        # a second pose: the bottle moved in front of the triangle
        pose = schauma.body2world_pose()
        pose[:3, 3] = f.body.body2world_pose()[:3, 3] + np.array([0.01, 0.0, -0.15], np.float32)
        schauma.set_body2world_pose(pose)
        for name in sorted(r):
            r[name].StartRendering()
            images.append(r[name].images())
        out.append(images)
    for a, b in zip(*out):
        assert np.array_equal(a[0], b[0])                      # depth, u16
        assert (a[1] is None and b[1] is None) or np.array_equal(a[1], b[1])  # silhouette ids
        assert a[2:] == b[2:]                                   # corner, scale, visibility
    assert (out[0][0][0] < 65535).sum() > 1000  # something was drawn


def test_tracking_step_with_all_branches_matches_oracle():
    """Region + Depth modality with region checking, silhouette checking and modelled occlusions
    through a whole tracking step (renderings refreshed before every correspondence search and
    for the histogram update); reference summation order -> bit-identical poses and states"""
    import ctypes as C
    import os
    res = []
    hip_fused, hip_one, hip_substep = util.open_hip(), util.open_hip(), util.open_hip()
    hip_fused.call("set_fused_step", 2)    # one launch per correspondence search (+ the renderers), state written back
    hip_one.call("set_fused_step", 2)      # the same with ONE workgroup for the object (M3T_HIP_NO_SPLIT)
    hip_substep.call("set_fused_step", 0)  # one launch per sub-step
    for api in (hip_fused, hip_one, hip_substep, util.open_oracle()):
        f, schauma, r = _scene(api, 200)
        f.region.ModelOcclusions(r["color_depth"])
        f.region.UseRegionChecking(r["color_sil"])
        f.depth.ModelOcclusions(r["depth_depth"])
        f.depth.UseSilhouetteChecking(r["depth_sil"])
        assert f.tracker.StartModalities(0)
        if api is hip_one:
            os.environ["M3T_HIP_NO_SPLIT"] = "1"
        try:
            assert f.tracker.ExecuteTrackingStep(0)
        finally:
            os.environ.pop("M3T_HIP_NO_SPLIT", None)
        lines = f.region.data_lines()
        points = f.depth.data_points()
        res.append((f.body.body2world_pose(), lines["valid"].copy(), points["valid"].copy(), f.region.histograms()))
        if api in (hip_fused, hip_one):
            name = C.create_string_buffer(64)
            api.call("get_step_kernel", name, 64)
            # one object: several workgroups share it (the split kernel with the renderer-fed branches compiled in)
            assert name.value.decode() == ("tracking_step_split_render_kernel" if api is hip_fused else "tracking_step_kernel") \
                or (api is hip_one and name.value.decode() == "tracking_step_lds_kernel")
    (pa, la, qa, ha), one, (pc, lc, qc, hc), (pb, lb, qb, hb) = res
    assert np.array_equal(one[0], pb) and np.array_equal(one[1], lb) and np.array_equal(one[2], qb)
    assert np.array_equal(one[3][0], hb[0]) and np.array_equal(one[3][1], hb[1])
    assert np.array_equal(lc, lb) and np.array_equal(qc, qb) and np.array_equal(pc, pb)
    assert np.array_equal(hc[0], hb[0]) and np.array_equal(hc[1], hb[1])
    assert np.array_equal(la, lb) and np.array_equal(qa, qb)
    assert 0 < la.sum() < 179 and 0 < qa.sum() < 182  # the branches removed something
    assert np.array_equal(pa, pb)
    assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1], hb[1])
