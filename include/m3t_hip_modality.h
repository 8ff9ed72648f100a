// m3t_hip_modality.h — the adapter a maintainer drops into the M3T tree ("adapter mode"):
// HipRegionModality / HipDepthModality are m3t::Modality subclasses (include/m3t/modality.h:56-155) that forward the
// seven steps to libm3t_hip.so and copy the device-computed gradient / Hessian back into gradient_ / hessian_, so
// that the reference's unmodified Link, Optimizer and Tracker keep working (link.cpp:188-191 only reads those).
//
// All adapters of one tracker share a HipBatch: Tracker calls the modalities one after another
// (tracker.cpp:447-457,471-479,503-517); the first call of a (iteration, corr_iteration[, opt_iteration]) round
// pushes the host Bodies' poses (PrecalculatePoseVariables reads Body every time, region_modality.cpp:1000) and
// launches the step for ALL registered modalities, the later calls of the round only fetch their own result.
//
// Two modes.  Adapter mode (default): the device computes correspondences and g/H, the host's Optimizer solves (one
// launch per sub-step, one read-back per g/H round).  Device-optimisation mode (HipBatch::UseDeviceOptimization,
// SURVEY 8(b) row 1): the UNMODIFIED Tracker::ExecuteTrackingStep loop (tracker.cpp:344-364) still calls every
// sub-step, but the first CalculateCorrespondences(iteration, 0) of a step runs the whole loop nest as ONE launch
// (m3t_hip_execute_tracking_step) and writes the resulting poses into the host Bodies; every
// CalculateGradientAndHessian of that step then leaves gradient_ / hessian_ zero, so the host's Optimizer solves
// (0 + lambda) theta = 0 -> theta = 0 and its pose update is the identity (optimizer.cpp:144-167, link.cpp:205-241);
// CalculateResults writes the device pose into Body once more (the histogram update already rode in the launch).
//
// Needs the M3T headers (and through them Eigen / OpenCV); links -lm3t_hip.  In this repository it is compiled
// against the interface stubs under tests/cpp/m3t_stub/ (tests/test_cpp_adapter.py).
#ifndef M3T_HIP_MODALITY_H_
#define M3T_HIP_MODALITY_H_

#include <m3t/body.h>
#include <m3t/camera.h>
#include <m3t/depth_model.h>
#include <m3t/modality.h>
#include <m3t/region_model.h>

#include <algorithm>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "m3t_hip.h"

namespace m3t_hip_adapter {

struct HipBatch {
  m3t_hip_context* ctx = nullptr;
  std::vector<std::pair<int /*device body id*/, std::shared_ptr<m3t::Body>>> bodies;

  explicit HipBatch(int device_id = 0) {
    if (m3t_hip_create(&ctx, device_id) < 0) {
      std::cerr << m3t_hip_last_error(nullptr) << std::endl;
      ctx = nullptr;
      return;
    }
    m3t_hip_set_fused_step(ctx, 0);  // the host drives the sub-steps one by one
  }
  ~HipBatch() {
    if (ctx) m3t_hip_destroy(ctx);
  }
  HipBatch(const HipBatch&) = delete;
  HipBatch& operator=(const HipBatch&) = delete;

  // one device body per host Body, shared by the modalities of that body
  int BodyId(const std::shared_ptr<m3t::Body>& body) {
    for (auto& b : bodies)
      if (b.second == body) return b.first;
    int id = m3t_hip_body_create(ctx, body->body2world_pose().data());
    if (id >= 0) bodies.push_back({id, body});
    return id;
  }
  // one device camera per host Camera, shared by the modalities that look through it
  struct SharedCamera {
    std::shared_ptr<m3t::Camera> camera;
    int id;
  };
  std::vector<SharedCamera> cameras;
  int ColorCameraId(const std::shared_ptr<m3t::ColorCamera>& camera) {
    for (auto& c : cameras)
      if (c.camera == camera) return c.id;
    const auto& in = camera->intrinsics();
    const m3t_intrinsics i{in.fu, in.fv, in.ppu, in.ppv, in.width, in.height};
    int id = m3t_hip_color_camera_create(ctx, &i, camera->world2camera_pose().data());
    if (id >= 0) cameras.push_back({camera, id});
    return id;
  }
  int DepthCameraId(const std::shared_ptr<m3t::DepthCamera>& camera) {
    for (auto& c : cameras)
      if (c.camera == camera) return c.id;
    const auto& in = camera->intrinsics();
    const m3t_intrinsics i{in.fu, in.fv, in.ppu, in.ppv, in.width, in.height};
    int id = m3t_hip_depth_camera_create(ctx, &i, camera->world2camera_pose().data(), camera->depth_scale());
    if (id >= 0) cameras.push_back({camera, id});
    return id;
  }
  // Camera::image() -> device for every registered camera.  Called once per round (Once() below de-duplicates
  // the calls of the modalities of one round); never keyed on the iteration index, which hosts repeat
  bool UploadAll() {
    bool ok = true;
    for (auto& c : cameras) {
      const cv::Mat& image = c.camera->image();  // BGR8 or u16; rows may be padded: data + step
      ok = m3t_hip_camera_upload(ctx, c.id, image.data, image.step) >= 0 && ok;
    }
    return ok;
  }
  void PushPoses() {
    for (auto& b : bodies) m3t_hip_body_set_body2world_pose(ctx, b.first, b.second->body2world_pose().data());
  }
  // device -> host Bodies (only needed by hosts that let the library optimise, "fast mode")
  bool PullPoses() {
    bool ok = true;
    for (auto& b : bodies) {
      m3t::Transform3fA pose;
      ok = m3t_hip_body_get_body2world_pose(ctx, b.first, pose.data()) >= 0 && ok;
      b.second->set_body2world_pose(pose);
    }
    return ok;
  }
  // "fast mode" (INTEGRATION.md §2): the library also optimises.  Register one rigid optimizer per body with the
  // device ids of its modalities (Optimizer::tikhonov_parameter_rotation / _translation, optimizer.h:135-136),
  // then Tracker::ExecuteTrackingStep is ExecuteTrackingStep() below.
  bool AddRigidOptimizer(const std::shared_ptr<m3t::Body>& body, const std::vector<int>& modality_ids,
                         float tikhonov_parameter_rotation = 1000.0f, float tikhonov_parameter_translation = 30000.0f) {
    const int rc = m3t_hip_optimizer_create_rigid(ctx, BodyId(body), int(modality_ids.size()), modality_ids.data(),
                                                  tikhonov_parameter_rotation, tikhonov_parameter_translation);
    if (rc < 0) std::cerr << m3t_hip_last_error(ctx) << std::endl;
    return rc >= 0;
  }
  // Device-optimisation mode for a host whose Tracker is left untouched (see the head of this file).  Every body needs
  // its AddRigidOptimizer first; n_corr_iterations / n_update_iterations are the Tracker's (tracker.h:231-232): the
  // device runs that many inside its one launch while the host's loop of the same length idles through.
  bool device_optimization = false;
  bool UseDeviceOptimization(int n_corr_iterations, int n_update_iterations) {
    device_optimization = Status(m3t_hip_set_fused_step(ctx, 1)) &&
                          Status(m3t_hip_tracker_set_iterations(ctx, n_corr_iterations, n_update_iterations));
    return device_optimization;
  }
  bool DoNotUseDeviceOptimization() {
    device_optimization = false;
    return Status(m3t_hip_set_fused_step(ctx, 0));
  }
  // StartModalities / ExecuteTrackingStep of tracker.cpp:344-364,430-445 for every registered body at once:
  // images up, poses up, the whole loop nest on the device, poses back into the host Bodies
  bool StartModalities(int iteration) {
    if (!UploadAll()) return Status(-1);
    PushPoses();
    return Status(m3t_hip_start_modalities(ctx, iteration));
  }
  bool ExecuteTrackingStep(int iteration, int n_corr_iterations, int n_update_iterations) {
    if (!UploadAll()) return Status(-1);
    PushPoses();
    return Status(m3t_hip_set_fused_step(ctx, 1)) &&
           Status(m3t_hip_tracker_set_iterations(ctx, n_corr_iterations, n_update_iterations)) &&
           Status(m3t_hip_execute_tracking_step(ctx, iteration)) && PullPoses();
  }
  bool Status(int rc) const {
    if (rc < 0) std::cerr << m3t_hip_last_error(ctx) << std::endl;
    return rc >= 0;
  }
  // One sub-step of the host's Tracker reaches every modality in turn (tracker.cpp:447-489); the first call of such a
  // round launches for the whole batch, the others of the round are served from it.  A round ends when its key
  // (the iteration indices) changes OR when a modality asks a second time: hosts legitimately repeat indices
  // (RBOTEvaluator::ResetBody -> StartModality(0, 0) after every loss, Refiner::RefinePoses on every call).
  struct Round {
    long key = 0;
    bool open = false, ok = true;
    std::vector<char> served;  // by device modality id
  };
  template <typename F>
  bool Once(Round* round, int modality_id, long key, bool with_images, F f) {
    if (modality_id >= int(round->served.size())) round->served.resize(size_t(modality_id) + 1, 0);
    if (!round->open || round->key != key || round->served[size_t(modality_id)]) {
      round->open = true;
      round->key = key;
      std::fill(round->served.begin(), round->served.end(), 0);
      round->ok = !with_images || UploadAll();
      PushPoses();
      const int rc = f();
      if (rc < 0) std::cerr << m3t_hip_last_error(ctx) << std::endl;
      round->ok = round->ok && rc >= 0;
    }
    round->served[size_t(modality_id)] = 1;
    return round->ok;
  }
  Round start_round, corr_round, gh_round, res_round, step_round;
  int n_modalities = 0;            // device modality ids are 0 .. n_modalities - 1
  std::vector<float> gh_cache;     // [n_modalities][6 + 36] of the last gradient / Hessian round
};

// Is `path` the sparse viewpoint model file of THIS model object and THIS body?  The acceptance test of
// Model::LoadModelParameters / LoadBodyData (model.cpp:218-284): type letter and version, the seven generation
// parameters (a file with MORE points per view passes, as in the reference), then the main body's geometry record.
// A .bin generated for another body or with other parameters would otherwise load silently.
inline bool ModelFileMatches(const std::filesystem::path& path, char model_type, int version_id, const m3t::Model& model) {
  std::ifstream ifs{path, std::ios::binary};
  if (!ifs.is_open()) return false;
  auto get = [&](auto* v) { ifs.read(reinterpret_cast<char*>(v), sizeof(*v)); return bool(ifs); };
  char type = 0;
  int version = 0, n_divides = 0, n_points = 0, image_size = 0;
  float sphere_radius = 0, max_radius_depth_offset = 0, stride_depth_offset = 0;
  bool use_random_seed = false;
  if (!(get(&type) && get(&version) && get(&sphere_radius) && get(&n_divides) && get(&n_points) &&
        get(&max_radius_depth_offset) && get(&stride_depth_offset) && get(&use_random_seed) && get(&image_size)))
    return false;
  if (type != model_type || version != version_id || sphere_radius != model.sphere_radius() ||
      n_divides != model.n_divides() || n_points < model.n_points() ||
      max_radius_depth_offset != model.max_radius_depth_offset() ||
      stride_depth_offset != model.stride_depth_offset() || use_random_seed != model.use_random_seed() ||
      image_size != model.image_size())
    return false;
  const m3t::Body& body = *model.body_ptr();
  std::string::size_type length = 0;
  if (!get(&length) || length > 4096) return false;
  std::string geometry_path(length, '\0');
  ifs.read(geometry_path.data(), std::streamsize(length));
  float unit = 0, diameter = 0;
  bool ccw = false, culling = false;
  float pose[16];
  if (!(ifs && get(&unit) && get(&ccw) && get(&culling) && get(&diameter))) return false;
  ifs.read(reinterpret_cast<char*>(pose), sizeof(pose));
  if (!ifs) return false;
  std::error_code ec;
  const bool same_file = std::filesystem::equivalent(geometry_path, body.geometry_path(), ec) ||
                         std::filesystem::path{geometry_path}.lexically_normal() == body.geometry_path().lexically_normal();
  return same_file && unit == body.geometry_unit_in_meter() && ccw == body.geometry_counterclockwise() &&
         culling == body.geometry_enable_culling() && diameter == body.maximum_body_diameter() &&
         std::memcmp(pose, body.geometry2body_pose().data(), sizeof(pose)) == 0;
}

// what both adapters do the same way
class HipModality : public m3t::Modality {
 public:
  bool StartModality(int iteration, int /*corr_iteration*/) override {
    if (!CheckSetUp()) return false;
    return batch_->Once(&batch_->start_round, id_, iteration, true,
                        [&] { return m3t_hip_start_modalities(batch_->ctx, iteration); });
  }
  bool CalculateCorrespondences(int iteration, int corr_iteration) override {
    if (!CheckSetUp()) return false;
    if (batch_->device_optimization) {
      // the first search of a step: the whole step on the device, poses into the host Bodies; the later searches
      // of the host's loop have nothing left to do
      if (corr_iteration > 0) return batch_->step_round.open && batch_->step_round.ok;
      return batch_->Once(&batch_->step_round, id_, iteration, true, [&] {
        const int rc = m3t_hip_execute_tracking_step(batch_->ctx, iteration);
        if (rc < 0) return rc;
        return batch_->PullPoses() ? 0 : -1;
      });
    }
    // (Tracker::UpdateCameras ran before the first search of a step: its images go up with that round)
    return batch_->Once(&batch_->corr_round, id_, iteration * 4096L + corr_iteration, corr_iteration == 0, [&] {
      return m3t_hip_calculate_correspondences(batch_->ctx, iteration, corr_iteration);
    });
  }
  bool CalculateGradientAndHessian(int iteration, int corr_iteration, int opt_iteration) override {
    if (!CheckSetUp()) return false;
    if (batch_->device_optimization) {  // g = 0, H = 0: the host's Optimizer finds theta = 0 (lambda > 0)
      std::fill(gradient_.data(), gradient_.data() + 6, 0.0f);
      std::fill(hessian_.data(), hessian_.data() + 36, 0.0f);
      return batch_->step_round.open && batch_->step_round.ok;
    }
    const bool ok = batch_->Once(&batch_->gh_round, id_, (iteration * 4096L + corr_iteration) * 4096L + opt_iteration, false, [&] {
      int rc = m3t_hip_calculate_gradient_and_hessian(batch_->ctx, iteration, corr_iteration, opt_iteration);
      if (rc < 0) return rc;
      // ONE read-back for the whole batch per round; the modalities of the round copy their 42 floats from it
      batch_->gh_cache.resize(size_t(batch_->n_modalities) * 42);
      return m3t_hip_modalities_get_gradient_hessian(batch_->ctx, batch_->gh_cache.data(), batch_->n_modalities);
    });
    if (!ok || size_t(id_ + 1) * 42 > batch_->gh_cache.size()) return false;
    // column-major like Eigen: what the unmodified Link adds up (link.cpp:188-191)
    const float* gh = batch_->gh_cache.data() + size_t(id_) * 42;
    std::copy(gh, gh + 6, gradient_.data());
    std::copy(gh + 6, gh + 42, hessian_.data());
    return true;
  }
  bool CalculateResults(int iteration) override {
    if (!CheckSetUp()) return false;
    if (batch_->device_optimization)  // (the histogram update rode in the step's launch) the device pose, exactly
      return batch_->step_round.open && batch_->step_round.ok && batch_->PullPoses();
    return batch_->Once(&batch_->res_round, id_, iteration, false,
                        [&] { return m3t_hip_calculate_results(batch_->ctx, iteration); });
  }
  // modality.h:96-103.  Renderers and ColorHistograms of these modalities live in the library (C-ABI ids): it starts
  // the renderings before the sub-steps that read them and clears / initialises / updates shared histograms around
  // its modalities itself, so the host Tracker, which collects the objects to start and to clear from these getters
  // (tracker.cpp:738-800; AddPtrIfNameNotExists skips null), must find none.
  std::vector<std::shared_ptr<m3t::Renderer>> start_modality_renderer_ptrs() const override { return {}; }
  std::vector<std::shared_ptr<m3t::Renderer>> correspondence_renderer_ptrs() const override { return {}; }
  std::vector<std::shared_ptr<m3t::Renderer>> results_renderer_ptrs() const override { return {}; }
  std::shared_ptr<m3t::ColorHistograms> color_histograms_ptr() const override { return nullptr; }
  bool VisualizeCorrespondences(int) override { return true; }
  bool VisualizeOptimization(int) override { return true; }
  bool VisualizeResults(int) override { return true; }
  int device_id() const { return id_; }

 protected:
  HipModality(const std::string& name, const std::shared_ptr<m3t::Body>& body, std::shared_ptr<HipBatch> batch)
      : m3t::Modality{name, body}, batch_{std::move(batch)} {}
  bool CheckSetUp() const {
    if (!set_up_) std::cerr << "Set up modality " << name_ << " first" << std::endl;  // region_modality.cpp:1813-1819
    return set_up_;
  }
  std::shared_ptr<HipBatch> batch_;
  int body_id_ = -1, id_ = -1;
};

class HipRegionModality : public HipModality {
 public:
  // region_model_path: the .bin the reference's RegionModel wrote (or m3t_hip_region_model_generate's);
  // depth_camera: optional, switches measured occlusions on like RegionModality::MeasureOcclusions
  HipRegionModality(const std::string& name, const std::shared_ptr<m3t::Body>& body,
                    std::shared_ptr<m3t::ColorCamera> color_camera, std::filesystem::path region_model_path,
                    std::shared_ptr<HipBatch> batch, const m3t_region_modality_params& params,
                    std::shared_ptr<m3t::DepthCamera> depth_camera = nullptr)
      : HipModality{name, body, std::move(batch)},
        color_camera_{std::move(color_camera)},
        depth_camera_{std::move(depth_camera)},
        model_path_{std::move(region_model_path)},
        params_{params} {}
  // The reference's signature (region_modality.h:169-172) with the batch and the parameter block appended: the host's
  // own RegionModel object.  Its views are private, so the adapter reads the file that model wrote or loaded in
  // its SetUp (model_path()), after checking that the file belongs to this model and body (ModelFileMatches).
  HipRegionModality(const std::string& name, const std::shared_ptr<m3t::Body>& body,
                    std::shared_ptr<m3t::ColorCamera> color_camera, std::shared_ptr<m3t::RegionModel> region_model,
                    std::shared_ptr<HipBatch> batch, const m3t_region_modality_params& params,
                    std::shared_ptr<m3t::DepthCamera> depth_camera = nullptr)
      : HipModality{name, body, std::move(batch)},
        color_camera_{std::move(color_camera)},
        depth_camera_{std::move(depth_camera)},
        region_model_{std::move(region_model)},
        params_{params} {}
  std::shared_ptr<m3t::Model> model_ptr() const override { return region_model_; }

  // RegionModality::ModelOcclusions / UseRegionChecking / UseSharedColorHistograms and their DoNot... twins
  // (region_modality.cpp:168-179, 230-267) with the library's object ids in place of the host's OpenGL renderers and
  // ColorHistograms: m3t_hip_focused_depth_renderer_create / m3t_hip_focused_silhouette_renderer_create /
  // m3t_hip_color_histograms_create on batch->ctx.  Like the reference's setters they take effect with the next SetUp.
  void ModelOcclusions(int depth_renderer_id) { depth_renderer_id_ = depth_renderer_id; set_up_ = false; }
  void DoNotModelOcclusions() { depth_renderer_id_ = -1; set_up_ = false; }
  void UseRegionChecking(int silhouette_renderer_id) { silhouette_renderer_id_ = silhouette_renderer_id; set_up_ = false; }
  void DoNotUseRegionChecking() { silhouette_renderer_id_ = -1; set_up_ = false; }
  void UseSharedColorHistograms(int color_histograms_id) { color_histograms_id_ = color_histograms_id; set_up_ = false; }
  void DoNotUseSharedColorHistograms() { color_histograms_id_ = -1; set_up_ = false; }

  bool SetUp() override {
    set_up_ = false;
    if (!batch_ || !batch_->ctx) return false;
    if (region_model_) {
      if (!region_model_->set_up()) {  // region_modality.cpp:37-40
        std::cerr << "Region model " << region_model_->name() << " was not set up" << std::endl;
        return false;
      }
      model_path_ = region_model_->model_path();
      if (!ModelFileMatches(model_path_, 'r', 10, *region_model_)) {
        std::cerr << "Model file " << model_path_ << " was not generated for region model " << region_model_->name()
                  << " and body " << region_model_->body_ptr()->name() << std::endl;
        return false;
      }
    }
    color_id_ = batch_->ColorCameraId(color_camera_);
    if (depth_camera_) {
      depth_id_ = batch_->DepthCameraId(depth_camera_);
      params_.measure_occlusions = 1;
    }
    model_id_ = m3t_hip_region_model_load(batch_->ctx, model_path_.string().c_str());
    body_id_ = batch_->BodyId(body_ptr_);
    if (color_id_ >= 0 && model_id_ >= 0 && body_id_ >= 0 && (!depth_camera_ || depth_id_ >= 0))
      id_ = m3t_hip_region_modality_create(batch_->ctx, &params_, body_id_, color_id_, model_id_, depth_id_);
    set_up_ = id_ >= 0;
    if (set_up_ && depth_renderer_id_ >= 0)
      set_up_ = m3t_hip_region_modality_model_occlusions(batch_->ctx, id_, depth_renderer_id_) >= 0;
    if (set_up_ && silhouette_renderer_id_ >= 0)
      set_up_ = m3t_hip_region_modality_use_region_checking(batch_->ctx, id_, silhouette_renderer_id_) >= 0;
    if (set_up_ && color_histograms_id_ >= 0)
      set_up_ = m3t_hip_region_modality_use_shared_color_histograms(batch_->ctx, id_, color_histograms_id_) >= 0;
    if (set_up_) batch_->n_modalities = std::max(batch_->n_modalities, id_ + 1);
    if (!set_up_) std::cerr << m3t_hip_last_error(batch_->ctx) << std::endl;
    return set_up_;
  }
  std::vector<std::shared_ptr<m3t::Camera>> camera_ptrs() const override {
    if (depth_camera_) return {color_camera_, depth_camera_};
    return {color_camera_};
  }

 private:
  std::shared_ptr<m3t::ColorCamera> color_camera_;
  std::shared_ptr<m3t::DepthCamera> depth_camera_;
  std::shared_ptr<m3t::RegionModel> region_model_;
  std::filesystem::path model_path_;
  m3t_region_modality_params params_;
  int color_id_ = -1, depth_id_ = -1, model_id_ = -1;
  int depth_renderer_id_ = -1, silhouette_renderer_id_ = -1, color_histograms_id_ = -1;
};

class HipDepthModality : public HipModality {
 public:
  HipDepthModality(const std::string& name, const std::shared_ptr<m3t::Body>& body,
                   std::shared_ptr<m3t::DepthCamera> depth_camera, std::filesystem::path depth_model_path,
                   std::shared_ptr<HipBatch> batch, const m3t_depth_modality_params& params)
      : HipModality{name, body, std::move(batch)},
        depth_camera_{std::move(depth_camera)},
        model_path_{std::move(depth_model_path)},
        params_{params} {}
  // depth_modality.h:110-113 with the batch and the parameter block appended (see HipRegionModality)
  HipDepthModality(const std::string& name, const std::shared_ptr<m3t::Body>& body,
                   std::shared_ptr<m3t::DepthCamera> depth_camera, std::shared_ptr<m3t::DepthModel> depth_model,
                   std::shared_ptr<HipBatch> batch, const m3t_depth_modality_params& params)
      : HipModality{name, body, std::move(batch)},
        depth_camera_{std::move(depth_camera)},
        depth_model_{std::move(depth_model)},
        params_{params} {}
  std::shared_ptr<m3t::Model> model_ptr() const override { return depth_model_; }

  // DepthModality::ModelOcclusions / UseSilhouetteChecking (depth_modality.cpp:128-161) with the library's renderer ids
  void ModelOcclusions(int depth_renderer_id) { depth_renderer_id_ = depth_renderer_id; set_up_ = false; }
  void DoNotModelOcclusions() { depth_renderer_id_ = -1; set_up_ = false; }
  void UseSilhouetteChecking(int silhouette_renderer_id) { silhouette_renderer_id_ = silhouette_renderer_id; set_up_ = false; }
  void DoNotUseSilhouetteChecking() { silhouette_renderer_id_ = -1; set_up_ = false; }

  bool SetUp() override {
    set_up_ = false;
    if (!batch_ || !batch_->ctx) return false;
    if (depth_model_) {
      if (!depth_model_->set_up()) {
        std::cerr << "Depth model " << depth_model_->name() << " was not set up" << std::endl;
        return false;
      }
      model_path_ = depth_model_->model_path();
      if (!ModelFileMatches(model_path_, 'd', 9, *depth_model_)) {
        std::cerr << "Model file " << model_path_ << " was not generated for depth model " << depth_model_->name()
                  << " and body " << depth_model_->body_ptr()->name() << std::endl;
        return false;
      }
    }
    depth_id_ = batch_->DepthCameraId(depth_camera_);
    model_id_ = m3t_hip_depth_model_load(batch_->ctx, model_path_.string().c_str());
    body_id_ = batch_->BodyId(body_ptr_);
    if (depth_id_ >= 0 && model_id_ >= 0 && body_id_ >= 0)
      id_ = m3t_hip_depth_modality_create(batch_->ctx, &params_, body_id_, depth_id_, model_id_);
    set_up_ = id_ >= 0;
    if (set_up_ && depth_renderer_id_ >= 0)
      set_up_ = m3t_hip_depth_modality_model_occlusions(batch_->ctx, id_, depth_renderer_id_) >= 0;
    if (set_up_ && silhouette_renderer_id_ >= 0)
      set_up_ = m3t_hip_depth_modality_use_silhouette_checking(batch_->ctx, id_, silhouette_renderer_id_) >= 0;
    if (set_up_) batch_->n_modalities = std::max(batch_->n_modalities, id_ + 1);
    if (!set_up_) std::cerr << m3t_hip_last_error(batch_->ctx) << std::endl;
    return set_up_;
  }
  std::vector<std::shared_ptr<m3t::Camera>> camera_ptrs() const override { return {depth_camera_}; }

 private:
  std::shared_ptr<m3t::DepthCamera> depth_camera_;
  std::shared_ptr<m3t::DepthModel> depth_model_;
  std::filesystem::path model_path_;
  m3t_depth_modality_params params_;
  int depth_id_ = -1, model_id_ = -1;
  int depth_renderer_id_ = -1, silhouette_renderer_id_ = -1;
};

}  // namespace m3t_hip_adapter

#endif  // M3T_HIP_MODALITY_H_
