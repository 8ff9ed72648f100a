"""Copies the reference's own golden vectors / fixtures for the hot path into
tests/golden/reference/ (run once in the build container, where /root/reference
is mounted; the GPU box has no /root/reference).  Data files only — no sources.

Provenance (all under /root/reference/M3T/data/):
  model_test/{region,depth}_model.bin      RegionModelTest/DepthModelTest goldens (test/model_test.cpp:165-184)
  model_test/depth_model_occlusion.bin, multi_region_model_{fixed,movable,same}.bin   the ValidationRule* goldens of
                                           models with associated bodies (test/model_test.cpp:232-320,508-538)
  modality_test/*_{gradient,hessian}.txt   modality g/H goldens (test/modality_test.cpp:280-316,534-550)
  modality_test/region_modality.png        lines-correspondence visualisation golden (test/modality_test.cpp:180-193)
  modality_test/*measured_occlusions.png, depth_modality.png   visualisation goldens with / without measured
                                           occlusions (test/modality_test.cpp:222-248,433-456,486-502)
  optimizer_test/triangle_pose.txt         OptimizerTest.Optimize golden (test/optimizer_test.cpp:97-105)
  tracker_test/, refiner_test/ pose goldens (test/tracker_test.cpp:164-195)
  _sequence/{color,depth}_camera_image_20{0,1}.png + yaml   the fixture frames (test/common_test.cpp:96-118)
  _body/triangle.obj, schauma.obj + yaml   fixture meshes
  renderer_test/focused_{depth,silhouette}_image.png   FocusedSilhouetteRendererTest / FocusedBasicDepthRendererTest
                                           goldens (test/renderer_test.cpp:301-323,892-902)
  pen_paper_demo/*.yaml                    the reference's demo configuration (examples/run_pen_paper_demo.cpp): a larger
                                           file for the YAML readers (unquoted names, comments, flow sequences)
  detector_test/detector_triangle_pose.txt StaticDetectorTest.DetectPose golden (test/detector_test.cpp:77-85)
  tracker_test/tracker_config.yaml, _body/triangle_{region,depth}_model.yaml   the generator configuration of
                                           TrackerTest.OptimizePoseMatrixGeneratorSetUp (test/tracker_test.cpp:182-195)
"""
import os
import shutil

SRC = "/root/reference/M3T/data"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference")
FILES = [
    "model_test/region_model.bin", "model_test/depth_model.bin", "model_test/depth_model_occlusion.bin",
    "model_test/multi_region_model_fixed.bin", "model_test/multi_region_model_movable.bin",
    "model_test/multi_region_model_same.bin", "model_test/region_model.yaml",
    "model_test/depth_model.yaml",
    "modality_test/region_modality_global_gradient.txt", "modality_test/region_modality_global_hessian.txt",
    "modality_test/region_modality_local_gradient.txt", "modality_test/region_modality_local_hessian.txt",
    "modality_test/depth_modality_gradient.txt", "modality_test/depth_modality_hessian.txt",
    "modality_test/region_modality.yaml", "modality_test/depth_modality.yaml", "modality_test/region_modality.png",
    "modality_test/region_modality_measured_occlusions.png", "modality_test/region_modality_depth_measured_occlusions.png",
    "modality_test/depth_modality.png", "modality_test/depth_modality_measured_occlusions.png",
    "modality_test/region_modality_region_checking.png", "modality_test/region_modality_silhouette_region_checking.png",
    "modality_test/region_modality_modeled_occlusions.png", "modality_test/region_modality_depth_modeled_occlusions.png",
    "modality_test/depth_modality_silhouette_checking.png", "modality_test/depth_modality_silhouette_silhouette_checking.png",
    "modality_test/depth_modality_modeled_occlusions.png", "modality_test/depth_modality_depth_modeled_occlusions.png",
    "optimizer_test/triangle_pose.txt", "optimizer_test/optimizer.yaml",
    "tracker_test/triangle_pose.txt", "tracker_test/tracker.yaml", "refiner_test/triangle_pose.txt",
    "_sequence/color_camera_image_200.png", "_sequence/color_camera_image_201.png",
    "_sequence/depth_camera_image_200.png", "_sequence/depth_camera_image_201.png",
    "_sequence/color_camera.yaml", "_sequence/depth_camera.yaml",
    "_body/triangle.obj", "_body/triangle.yaml", "_body/schauma.yaml", "_body/schauma.obj",
    "_body/triangle_static_detector.yaml", "color_histograms_test/color_histograms.yaml",
    "tracker_test/tracker_config.yaml", "_body/triangle_region_model.yaml", "_body/triangle_depth_model.yaml",
    "renderer_test/focused_depth_image.png", "renderer_test/focused_silhouette_image.png",
    "detector_test/detector_triangle_pose.txt",
    "pen_paper_demo/config.yaml", "pen_paper_demo/stabilo_region_modality.yaml", "pen_paper_demo/paper_detector.yaml",
]

if __name__ == "__main__":
    for f in FILES:
        d = os.path.join(DST, f)
        os.makedirs(os.path.dirname(d), exist_ok=True)
        shutil.copyfile(os.path.join(SRC, f), d)
        os.chmod(d, 0o644)
        print("copied", f, os.path.getsize(d))
