"""The renderer-fed branches against the reference's goldens
(test/modality_test.cpp:195-220,250-278,458-484,504-533): region checking and modelled occlusions of
the RegionModality, silhouette checking and modelled occlusions of the DepthModality, on the
triangle occluded by the schauma bottle.  What is compared: the silhouette rendering pixel for pixel
(its golden image is the rendering itself), and the set of lines / points that survive, read off
the marks drawn into the full-size and focused visualisation goldens.  The grey depth backgrounds
are not compared (8-bit truncation of a depth that is reproduced to one 16-bit step)."""
import numpy as np
import pytest

import golden_scene as gs
import util
from util import host

LINE = (24, 184, 234)
POINT = (187, 117, 0)


def _lines_image_matches(f, name, n_lines):
    lines = f.modality.data_lines()
    lines = lines[lines["valid"] != 0]
    assert len(lines) == n_lines
    hf, hb = f.modality.histograms()
    vis = gs.render_lines_visualisation(f.image, hf, hb, lines, n_bins=16, scale=6, distribution_length=12)
    gold = gs.load_png("modality_test/" + name).astype(np.int32)
    assert int((np.abs(vis - gold).max(axis=2) > 0).sum()) == 0
    return lines


def check_region_checking(api):
    f = gs.RegionFixture(api)
    geometry, _ = gs.fixture_renderer_geometry(api, f.body)
    renderer = host.FocusedSilhouetteRenderer(api, geometry, f.camera, id_type=1)  # IDType::REGION
    renderer.AddReferencedBody(f.body)
    f.modality.UseRegionChecking(renderer)
    assert f.tracker.StartModalities(0) and f.tracker.CalculateCorrespondences(0, 0)
    lines = _lines_image_matches(f, "region_modality_region_checking.png", 132)
    _, silhouette, cu, cv, scale, n_visible = renderer.images()
    assert n_visible == 1
    gold = gs.load_png("modality_test/region_modality_silhouette_region_checking.png").astype(np.int32)
    marks = (gold == np.asarray(LINE)).all(axis=2)
    assert np.array_equal(silhouette.astype(np.int32)[~marks], gold[..., 0][~marks])  # the rendering itself
    assert np.array_equal(gs.focused_point_mask(200, lines["center_f_body"], gs.mtv.body2world(),
                                                gs.mtv.COLOR_INTRINSICS, cu, cv, scale), marks)


def check_region_modeled_occlusions(api):
    f = gs.RegionFixture(api, n_unoccluded_iterations=0)
    geometry, _ = gs.fixture_renderer_geometry(api, f.body)
    renderer = host.FocusedBasicDepthRenderer(api, geometry, f.camera)
    renderer.AddReferencedBody(f.body)
    f.modality.ModelOcclusions(renderer)
    assert f.tracker.StartModalities(0) and f.tracker.CalculateCorrespondences(0, 0)
    lines = _lines_image_matches(f, "region_modality_modeled_occlusions.png", 110)
    _, _, cu, cv, scale, _ = renderer.images()
    assert np.array_equal(gs.focused_point_mask(200, lines["center_f_body"], gs.mtv.body2world(),
                                                gs.mtv.COLOR_INTRINSICS, cu, cv, scale),
                          gs.golden_point_mask("modality_test/region_modality_depth_modeled_occlusions.png", LINE))


def check_depth_branches(api):
    for silhouette in (True, False):
        f = gs.DepthFixture(api, n_unoccluded_iterations=0)
        geometry, _ = gs.fixture_renderer_geometry(api, f.body)
        if silhouette:
            renderer = host.FocusedSilhouetteRenderer(api, geometry, f.camera, id_type=0)  # IDType::BODY
            renderer.AddReferencedBody(f.body)
            f.modality.UseSilhouetteChecking(renderer)
            names = ("depth_modality_silhouette_checking.png", "depth_modality_silhouette_silhouette_checking.png")
            n_points, mark = 146, LINE
        else:
            renderer = host.FocusedBasicDepthRenderer(api, geometry, f.camera)
            renderer.AddReferencedBody(f.body)
            f.modality.ModelOcclusions(renderer)
            names = ("depth_modality_modeled_occlusions.png", "depth_modality_depth_modeled_occlusions.png")
            n_points, mark = 116, POINT
        assert f.tracker.CalculateCorrespondences(0, 0)
        points = f.modality.data_points()
        points = points[points["valid"] != 0]
        assert len(points) == n_points
        assert np.array_equal(gs.point_mask((480, 848), points["center_f_body"], f.body2camera,
                                            gs.mtv.DEPTH_INTRINSICS),
                              gs.golden_point_mask("modality_test/" + names[0], POINT))
        _, sil, cu, cv, scale, _ = renderer.images()
        gold = gs.load_png("modality_test/" + names[1]).astype(np.int32)
        marks = (gold == np.asarray(mark)).all(axis=2)
        assert np.array_equal(gs.focused_point_mask(200, points["center_f_body"], f.body2camera,
                                                    gs.mtv.DEPTH_INTRINSICS, cu, cv, scale), marks)
        if silhouette:  # ids 150 (triangle) and 50 (bottle)
            assert np.array_equal(sil.astype(np.int32)[~marks], gold[..., 0][~marks])


def check_focused_renderer_images(api):
    """FocusedSilhouetteRendererTest.TestSilhouetteImage / TestDepthImage and FocusedBasicDepthRendererTest.
    TestDepthImage (test/renderer_test.cpp:152-193,301-323,748-778,892-902): triangle + bottle seen by a 640 x 480
    camera 1 cm off the origin, focused on the triangle at 200 x 200, z range 0.1-2 m.  The reference's criterion
    is CompareToLoadedImage(..., 0, 10): at most 10 pixels further than 1 from the golden.  Met with no such pixel:
    the silhouette (body ids) is identical, the 16-bit depth buffer differs by one step on < 1 % of the pixels
    (the last bit of the hardware's depth interpolation)."""
    body = host.Body(api, gs.mtv.body2world())
    geometry, _ = gs.fixture_renderer_geometry(api, body)
    world2camera = np.eye(4, dtype=np.float32)
    world2camera[0, 3] = 0.01
    camera = host.ColorCamera(api, 698.128, 698.617, 478.459, 274.426, 640, 480, world2camera_pose=world2camera)
    silhouette_renderer = host.FocusedSilhouetteRenderer(api, geometry, camera, id_type=0, image_size=200, z_min=0.1,
                                                        z_max=2.0)
    silhouette_renderer.AddReferencedBody(body)
    depth_renderer = host.FocusedBasicDepthRenderer(api, geometry, camera, image_size=200, z_min=0.1, z_max=2.0)
    depth_renderer.AddReferencedBody(body)
    assert silhouette_renderer.StartRendering() and depth_renderer.StartRendering()
    depth, silhouette, _, _, _, n_visible = silhouette_renderer.images()
    assert n_visible == 1
    gold_depth = gs.load_png("renderer_test/focused_depth_image.png")
    gold_silhouette = gs.load_png("renderer_test/focused_silhouette_image.png")
    assert gold_depth.dtype == np.uint16 and gold_depth.shape == (200, 200)
    assert np.array_equal(silhouette, gold_silhouette)
    assert set(np.unique(silhouette)) == {0, 50, 150}  # background, bottle, triangle (IDType::BODY)
    difference = np.abs(depth.astype(np.int64) - gold_depth.astype(np.int64))
    assert int((difference > 1).sum()) == 0            # the reference allows 10
    assert int((difference > 0).sum()) < 400            # of 14 889 covered pixels
    assert np.array_equal(depth_renderer.images()[0], depth)  # both renderers fill the same depth buffer
    return depth, silhouette


def test_oracle_focused_renderer_images():
    check_focused_renderer_images(util.open_oracle())


@pytest.mark.gpu
def test_hip_focused_renderer_images():
    depth, silhouette = check_focused_renderer_images(util.open_hip())
    oracle_depth, oracle_silhouette = check_focused_renderer_images(util.open_oracle())
    assert np.array_equal(depth, oracle_depth) and np.array_equal(silhouette, oracle_silhouette)


@pytest.mark.parametrize("check", [check_region_checking, check_region_modeled_occlusions, check_depth_branches])
def test_oracle_renderer_goldens(check):
    check(util.open_oracle())


@pytest.mark.gpu
@pytest.mark.parametrize("check", [check_region_checking, check_region_modeled_occlusions, check_depth_branches])
def test_hip_renderer_goldens(check):
    check(util.open_hip())
