// C++ host driving the hot path through include/m3t_hip.hpp (the C++ mirror of the reference's
// Tracker / Modality surface).  Input: a directory written by tests/test_cpp_host.py:
//   scene.txt : n_objects n_frames width height fu fv ppu ppv n_views n_points
//   model_<i>.bin (points, orientations, contour lengths), start_<i>.bin (pose), frame_<i>_<k>.bin (BGR8)
// Output: the final pose of every object, one line of 16 floats (hex-exact via %a) each.
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "m3t_hip.hpp"

static std::vector<char> ReadAll(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  std::string dir = argv[1];
  int n_objects, n_frames, n_views, n_points;
  m3t_intrinsics intr;
  {
    std::ifstream f(dir + "/scene.txt");
    f >> n_objects >> n_frames >> intr.width >> intr.height >> intr.fu >> intr.fv >> intr.ppu >> intr.ppv >> n_views >>
        n_points;
  }
  auto ctx = std::make_shared<m3t_hip::Context>(0);
  m3t_region_modality_params params;
  m3t_region_modality_params_default(&params);
  // RBOT parameter set (examples/evaluate_rbot_dataset.cpp:25-44)
  params.function_amplitude = 0.36f;
  params.function_slope = 0.0f;
  params.n_scales = 4;
  params.scales[0] = 5; params.scales[1] = 2; params.scales[2] = 2; params.scales[3] = 1;
  params.n_standard_deviations = 4;
  params.standard_deviations[0] = 20.0f; params.standard_deviations[1] = 7.0f;
  params.standard_deviations[2] = 3.0f; params.standard_deviations[3] = 1.5f;
  params.n_histogram_bins = 32;
  params.n_unoccluded_iterations = 0;

  std::vector<std::unique_ptr<m3t_hip::Body>> bodies;
  std::vector<std::unique_ptr<m3t_hip::ColorCamera>> cameras;
  std::vector<std::unique_ptr<m3t_hip::RegionModel>> models;
  std::vector<std::unique_ptr<m3t_hip::RegionModality>> modalities;
  std::vector<std::unique_ptr<m3t_hip::Optimizer>> optimizers;
  for (int i = 0; i < n_objects; ++i) {
    auto mb = ReadAll(dir + "/model_" + std::to_string(i) + ".bin");
    const float* mf = reinterpret_cast<const float*>(mb.data());
    m3t_region_model_desc desc{};
    desc.n_views = n_views;
    desc.n_points = n_points;
    desc.data_points = mf;
    desc.orientations = mf + size_t(n_views) * n_points * M3T_REGION_POINT_FLOATS;
    desc.contour_lengths = desc.orientations + size_t(n_views) * 3;
    desc.stride_depth_offset = 0.002f;
    desc.max_radius_depth_offset = 0.05f;
    models.push_back(std::make_unique<m3t_hip::RegionModel>(ctx, desc));
    auto sb = ReadAll(dir + "/start_" + std::to_string(i) + ".bin");
    m3t_hip::Pose start;
    std::copy(reinterpret_cast<const float*>(sb.data()), reinterpret_cast<const float*>(sb.data()) + 16, start.begin());
    bodies.push_back(std::make_unique<m3t_hip::Body>(ctx, start));
    cameras.push_back(std::make_unique<m3t_hip::ColorCamera>(ctx, intr));
    modalities.push_back(
        std::make_unique<m3t_hip::RegionModality>(ctx, *bodies.back(), *cameras.back(), *models.back(), params));
    optimizers.push_back(std::make_unique<m3t_hip::Optimizer>(
        ctx, *bodies.back(), std::vector<const m3t_hip::Modality*>{modalities.back().get()}, 1000.0f, 30000.0f));
  }
  m3t_hip::Tracker tracker(ctx, 7, 2);
  if (tracker.StartModalities(0)) return 3;  // must fail: no image yet ("Set up ... first")
  for (int k = 0; k < n_frames; ++k) {
    for (int i = 0; i < n_objects; ++i) {
      auto fb = ReadAll(dir + "/frame_" + std::to_string(i) + "_" + std::to_string(k) + ".bin");
      if (!cameras[i]->UpdateImage(fb.data(), size_t(intr.width) * 3)) return 4;
    }
    if (k == 0 && !tracker.StartModalities(0)) return 5;
    if (!tracker.ExecuteTrackingStep(k)) return 6;
  }
  for (int i = 0; i < n_objects; ++i) {
    m3t_hip::Pose p = bodies[i]->body2world_pose();
    for (float v : p) std::printf("%a ", v);
    std::printf("\n");
  }
  return 0;
}
