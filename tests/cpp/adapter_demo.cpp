// A host that only knows m3t::Modality drives include/m3t_hip_modality.h the way Tracker does
// (tracker.cpp:430-517): every step for every modality in turn.  Built against the interface stubs of
// tests/cpp/m3t_stub/ by tests/test_cpp_adapter.py — once with the C-ABI names mapped onto the CPU oracle (runs
// anywhere), once against libm3t_hip.so (GPU).
//   adapter_demo SCENE_DIR [adapter-only]     SCENE_DIR/scene.txt: color intrinsics, depth intrinsics + scale, 16 floats depth
//                              world2camera (column-major), 16 floats body2world; region.bin, depth.bin, color.raw,
//                              depth.raw next to it
// Prints per modality its name, 6 gradient and 36 Hessian floats (hex) after one correspondence search.
#include <cstdio>
#include <fstream>

#include "m3t_hip_modality.h"

namespace {
std::vector<unsigned char> ReadFile(const std::string& path) {
  std::ifstream ifs(path, std::ios::binary);
  return std::vector<unsigned char>((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
}
template <typename BASE>
class RawFileCamera : public BASE {
 public:
  RawFileCamera(const std::string& name, const std::string& path, const m3t::Intrinsics& intrinsics, int bytes_per_pixel)
      : BASE{name}, path_{path}, bytes_per_pixel_{bytes_per_pixel} {
    this->intrinsics_ = intrinsics;
  }
  bool SetUp() override { return UpdateImage(true); }
  bool UpdateImage(bool) override {
    pixels_ = ReadFile(path_);
    const size_t row = size_t(this->intrinsics_.width) * size_t(bytes_per_pixel_);
    if (pixels_.size() != row * size_t(this->intrinsics_.height)) return false;
    this->image_.data = pixels_.data();
    this->image_.step = row;
    this->image_.rows = this->intrinsics_.height;
    this->image_.cols = this->intrinsics_.width;
    return true;
  }
  void set_world2camera_pose(const m3t::Transform3fA& pose) { this->world2camera_pose_ = pose; }
  void set_depth_scale(float s) { SetScale(s, this); }

 private:
  static void SetScale(float s, m3t::DepthCamera* c) {
    struct Access : m3t::DepthCamera {
      using m3t::DepthCamera::depth_scale_;
    };
    c->*(&Access::depth_scale_) = s;
  }
  static void SetScale(float, m3t::ColorCamera*) {}
  std::string path_;
  int bytes_per_pixel_;
  std::vector<unsigned char> pixels_;
};

// The reference's host side above the modalities, reduced to one free body, for the device-optimisation mode: the
// loop nest is tracker.cpp:344-364 / :447-517 as written there; the optimizer is Link::CalculateGradientAndHessian
// (link.cpp:184-193: sum over the modalities), Optimizer::CalculateOptimization (optimizer.cpp:144-167: b = g,
// A = -H + diag(lambda), solve, NaN guard) and Link::UpdatePoses (link.cpp:205-241: T <- T [R(theta_r) | theta_t]).
struct HostOptimizer {
  std::shared_ptr<m3t::Body> body;
  std::vector<std::shared_ptr<m3t::Modality>> modalities;
  float tikhonov_rotation = 1000.0f, tikhonov_translation = 30000.0f;
  double largest_theta = 0.0;  // what the test looks at: the host's own solve found nothing left to do
  bool CalculateOptimization(int, int, int) {
    double a[6][7] = {};
    for (auto& m : modalities) {
      for (int i = 0; i < 6; ++i) {
        a[i][6] += double(m->gradient().v[size_t(i)]);
        for (int j = 0; j < 6; ++j) a[i][j] -= double(m->hessian().v[size_t(j) * 6 + size_t(i)]);
      }
    }
    for (int i = 0; i < 6; ++i) a[i][i] += double(i < 3 ? tikhonov_rotation : tikhonov_translation);
    for (int k = 0; k < 6; ++k) {  // (symmetric positive definite: no pivoting needed for this check)
      for (int i = k + 1; i < 6; ++i) {
        const double f = a[i][k] / a[k][k];
        for (int j = k; j < 7; ++j) a[i][j] -= f * a[k][j];
      }
    }
    double theta[6];
    for (int i = 5; i >= 0; --i) {
      double v = a[i][6];
      for (int j = i + 1; j < 6; ++j) v -= a[i][j] * theta[j];
      theta[i] = v / a[i][i];
    }
    for (double v : theta) {
      if (v != v) return true;  // optimizer.cpp:165-166
      largest_theta = std::max(largest_theta, v < 0 ? -v : v);
    }
    // T <- T * [expm(skew(theta_r)) | theta_t]; first-order rotation is enough for a test that expects theta = 0
    const m3t::Transform3fA t = body->body2world_pose();
    const float r[3][3] = {{1.0f, float(-theta[2]), float(theta[1])},
                           {float(theta[2]), 1.0f, float(-theta[0])},
                           {float(-theta[1]), float(theta[0]), 1.0f}};
    m3t::Transform3fA out = t;
    for (int c = 0; c < 3; ++c)
      for (int row = 0; row < 3; ++row)
        out.m[size_t(c) * 4 + size_t(row)] = t.m[size_t(row)] * r[0][c] + t.m[4 + size_t(row)] * r[1][c] + t.m[8 + size_t(row)] * r[2][c];
    for (int row = 0; row < 3; ++row)
      out.m[12 + size_t(row)] = t.m[size_t(row)] * float(theta[3]) + t.m[4 + size_t(row)] * float(theta[4]) +
                                t.m[8 + size_t(row)] * float(theta[5]) + t.m[12 + size_t(row)];
    body->set_body2world_pose(out);
    return true;
  }
};
struct HostTracker {  // tracker.cpp:344-364 with the sub-steps of :447-517 (no renderers, no shared histograms here)
  std::vector<std::shared_ptr<m3t::Modality>> modalities;
  std::vector<HostOptimizer*> optimizers;
  int n_corr_iterations = 7, n_update_iterations = 2;
  bool ExecuteTrackingStep(int iteration) {
    for (int corr_iteration = 0; corr_iteration < n_corr_iterations; ++corr_iteration) {
      int corr_save_idx = iteration * n_corr_iterations + corr_iteration;
      for (auto& m : modalities)
        if (!m->CalculateCorrespondences(iteration, corr_iteration)) return false;
      for (auto& m : modalities)
        if (!m->VisualizeCorrespondences(corr_save_idx)) return false;
      for (int update_iteration = 0; update_iteration < n_update_iterations; ++update_iteration) {
        int update_save_idx = corr_save_idx * n_update_iterations + update_iteration;
        for (auto& m : modalities)
          if (!m->CalculateGradientAndHessian(iteration, corr_iteration, update_iteration)) return false;
        for (auto* o : optimizers)
          if (!o->CalculateOptimization(iteration, corr_iteration, update_iteration)) return false;
        for (auto& m : modalities)
          if (!m->VisualizeOptimization(update_save_idx)) return false;
      }
    }
    for (auto& m : modalities)
      if (!m->CalculateResults(iteration)) return false;
    for (auto& m : modalities)
      if (!m->VisualizeResults(iteration)) return false;
    return true;
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  const std::string dir = argv[1];
  std::ifstream scene(dir + "/scene.txt");
  m3t::Intrinsics ci{}, di{};
  float depth_scale = 0.0f;
  m3t::Transform3fA depth_world2camera, body2world;
  scene >> ci.fu >> ci.fv >> ci.ppu >> ci.ppv >> ci.width >> ci.height;
  scene >> di.fu >> di.fv >> di.ppu >> di.ppv >> di.width >> di.height >> depth_scale;
  for (float& v : depth_world2camera.m) scene >> v;
  for (float& v : body2world.m) scene >> v;
  if (!scene) return 2;

  auto body = std::make_shared<m3t::Body>("triangle");
  body->set_body2world_pose(body2world);
  auto color = std::make_shared<RawFileCamera<m3t::ColorCamera>>("color_camera", dir + "/color.raw", ci, 3);
  auto depth = std::make_shared<RawFileCamera<m3t::DepthCamera>>("depth_camera", dir + "/depth.raw", di, 2);
  depth->set_world2camera_pose(depth_world2camera);
  depth->set_depth_scale(depth_scale);
  if (!color->SetUp() || !depth->SetUp()) return 3;

  using namespace m3t_hip_adapter;
  auto batch = std::make_shared<HipBatch>(0);
  if (!batch->ctx) return 4;
  m3t_region_modality_params rp;
  m3t_region_modality_params_default(&rp);
  m3t_depth_modality_params dp;
  m3t_depth_modality_params_default(&dp);
  dp.measure_occlusions = 1;
  std::vector<std::shared_ptr<m3t::Modality>> modalities{
      std::make_shared<HipRegionModality>("triangle_region_modality", body, color, dir + "/region.bin", batch, rp, depth),
      std::make_shared<HipDepthModality>("triangle_depth_modality", body, depth, dir + "/depth.bin", batch, dp)};

  if (modalities[0]->CalculateCorrespondences(0, 0)) return 5;  // "Set up modality ... first" (TestWithoutSetUp)
  for (auto& m : modalities)
    if (!m->SetUp()) return 6;
  if (batch->bodies.size() != 1 || batch->cameras.size() != 2) return 7;  // one device body, cameras shared
  for (auto& m : modalities)
    if (!m->StartModality(0, 0)) return 8;
  for (auto& m : modalities)
    if (!m->CalculateCorrespondences(0, 0)) return 9;
  for (auto& m : modalities)
    if (!m->CalculateGradientAndHessian(0, 0, 0)) return 10;
  for (auto& m : modalities) {
    std::printf("%s", m->name().c_str());
    for (float v : m->gradient().v) std::printf(" %a", double(v));
    for (float v : m->hessian().v) std::printf(" %a", double(v));
    std::printf("\n");
  }
  // the host moves the body (its own optimizer would): the next round pushes the new pose to the device
  m3t::Transform3fA moved = body->body2world_pose();
  moved.m[12] += 0.001f;
  body->set_body2world_pose(moved);
  for (auto& m : modalities)
    if (!m->CalculateGradientAndHessian(0, 0, 1)) return 11;
  std::printf("moved");
  for (float v : modalities[1]->gradient().v) std::printf(" %a", double(v));
  std::printf("\n");
  for (auto& m : modalities)
    if (!m->CalculateResults(0)) return 12;

  // hosts repeat iteration indices: RBOTEvaluator::ResetBody calls StartModality(0, 0) after every tracking loss and
  // starts over.  Same pose, same images, same indices must run again (not be served from the earlier round) and
  // reproduce the first gradient / Hessian
  body->set_body2world_pose(body2world);
  for (auto& m : modalities)
    if (!m->StartModality(0, 0)) return 17;
  for (auto& m : modalities)
    if (!m->CalculateCorrespondences(0, 0)) return 18;
  for (auto& m : modalities)
    if (!m->CalculateGradientAndHessian(0, 0, 0)) return 19;
  std::printf("again");
  for (float v : modalities[0]->gradient().v) std::printf(" %a", double(v));
  for (float v : modalities[0]->hessian().v) std::printf(" %a", double(v));
  std::printf("\n");

  if (argc > 2 && std::string(argv[2]) == "adapter-only") return 0;
  // fast mode on a second batch: the library optimises, the host Body receives the pose
  // ... built the way a maintainer's code builds it (region_modality.h:169-172, depth_modality.h:110-113): the host's
  // own model objects go into the constructors; the adapter reads the files they wrote / loaded after checking that
  // the files belong to these models and this body (Model::LoadModelParameters / LoadBodyData, model.cpp:218-284)
  auto fast_body = std::make_shared<m3t::Body>("triangle", "triangle.obj", 1.0f, true, true, m3t::Transform3fA{});
  fast_body->set_maximum_body_diameter(0.1f);
  fast_body->set_body2world_pose(body2world);
  auto fast = std::make_shared<HipBatch>(0);
  if (!fast->ctx) return 13;
  auto region_model = std::make_shared<m3t::RegionModel>("triangle_region_model", fast_body, dir + "/region.bin");
  auto depth_model = std::make_shared<m3t::DepthModel>("triangle_depth_model", fast_body, dir + "/depth.bin");
  {
    auto not_set_up = std::make_shared<HipRegionModality>("region", fast_body, color, region_model, fast, rp, depth);
    if (not_set_up->SetUp()) return 20;  // "Region model ... was not set up" (region_modality.cpp:37-40)
  }
  if (!region_model->SetUp() || !depth_model->SetUp()) return 21;
  {
    // a model object with other generation parameters, or for another body: the file on disk is not its file
    auto other = std::make_shared<m3t::RegionModel>("other", fast_body, dir + "/region.bin", 0.8f, 3);
    other->SetUp();
    if (std::make_shared<HipRegionModality>("region", fast_body, color, other, fast, rp, depth)->SetUp()) return 22;
    auto other_body = std::make_shared<m3t::Body>("bottle", "schauma.obj", 1.0f, true, true, m3t::Transform3fA{});
    other_body->set_maximum_body_diameter(0.1f);
    auto stale = std::make_shared<m3t::DepthModel>("stale", other_body, dir + "/depth.bin");
    stale->SetUp();
    if (std::make_shared<HipDepthModality>("depth", other_body, depth, stale, fast, dp)->SetUp()) return 23;
  }
  auto region = std::make_shared<HipRegionModality>("region", fast_body, color, region_model, fast, rp, depth);
  auto depth_modality = std::make_shared<HipDepthModality>("depth", fast_body, depth, depth_model, fast, dp);
  if (!region->SetUp() || !depth_modality->SetUp()) return 14;
  if (region->model_ptr() != region_model) return 24;
  if (!fast->AddRigidOptimizer(fast_body, {region->device_id(), depth_modality->device_id()})) return 15;
  if (!fast->StartModalities(0) || !fast->ExecuteTrackingStep(0, 7, 2)) return 16;
  std::printf("fast");
  for (float v : fast_body->body2world_pose().m) std::printf(" %a", double(v));
  std::printf("\n");

  // device-optimisation mode: the UNMODIFIED host loop (HostTracker = tracker.cpp:344-364) over the adapters; the
  // first correspondence search of the step runs the whole step on the device, the host's optimizer finds theta = 0
  auto fused_body = std::make_shared<m3t::Body>("triangle", "triangle.obj", 1.0f, true, true, m3t::Transform3fA{});
  fused_body->set_maximum_body_diameter(0.1f);
  fused_body->set_body2world_pose(body2world);
  auto fused = std::make_shared<HipBatch>(0);
  if (!fused->ctx) return 30;
  auto fused_region = std::make_shared<HipRegionModality>("region", fused_body, color, region_model, fused, rp, depth);
  auto fused_depth = std::make_shared<HipDepthModality>("depth", fused_body, depth, depth_model, fused, dp);
  if (!fused_region->SetUp() || !fused_depth->SetUp()) return 31;
  // the getters Tracker assembles its renderer / histogram lists from (tracker.cpp:738-800): nothing for the host to run
  if (!fused_region->start_modality_renderer_ptrs().empty() || !fused_region->correspondence_renderer_ptrs().empty() ||
      !fused_region->results_renderer_ptrs().empty() || fused_region->color_histograms_ptr() ||
      fused_depth->color_histograms_ptr())
    return 32;
  if (!fused->AddRigidOptimizer(fused_body, {fused_region->device_id(), fused_depth->device_id()})) return 33;
  HostOptimizer host_optimizer;
  host_optimizer.body = fused_body;
  host_optimizer.modalities = {fused_region, fused_depth};
  HostTracker host_tracker;
  host_tracker.modalities = {fused_region, fused_depth};
  host_tracker.optimizers = {&host_optimizer};
  if (!fused->UseDeviceOptimization(host_tracker.n_corr_iterations, host_tracker.n_update_iterations)) return 34;
  for (auto& m : host_tracker.modalities)
    if (!m->StartModality(0, 0)) return 35;
  if (!host_tracker.ExecuteTrackingStep(0)) return 36;
  if (host_optimizer.largest_theta != 0.0) return 37;  // 14 host solves, all of them no-ops
  for (float v : fused_region->hessian().v)
    if (v != 0.0f) return 38;
  m3t::Transform3fA device_pose;
  if (m3t_hip_body_get_body2world_pose(fused->ctx, fused->BodyId(fused_body), device_pose.data()) < 0) return 39;
  if (device_pose.m != fused_body->body2world_pose().m) return 40;
  std::printf("fused");
  for (float v : fused_body->body2world_pose().m) std::printf(" %a", double(v));
  std::printf("\n");
  // a region modality that names renderers / shared histograms the context does not have: SetUp fails loudly
  {
    auto bad = std::make_shared<HipRegionModality>("region", fused_body, color, region_model, fused, rp, depth);
    bad->UseRegionChecking(12345);
    if (bad->SetUp()) return 41;
    bad->DoNotUseRegionChecking();
    bad->UseSharedColorHistograms(m3t_hip_color_histograms_create(fused->ctx, 16, 0.2f, 0.2f));
    if (!bad->SetUp()) return 42;
  }
  return 0;
}
