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


@pytest.mark.parametrize("check", [check_region_checking, check_region_modeled_occlusions, check_depth_branches])
def test_oracle_renderer_goldens(check):
    check(util.open_oracle())


@pytest.mark.gpu
@pytest.mark.parametrize("check", [check_region_checking, check_region_modeled_occlusions, check_depth_branches])
def test_hip_renderer_goldens(check):
    check(util.open_hip())
