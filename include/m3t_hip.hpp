// m3t_hip.hpp — header-only C++ host mirror of the reference's object graph for the hot
// path, on top of the C-ABI (m3t_hip.h).  Class and method names follow M3T
// (Body, ColorCamera, DepthCamera, RegionModel, DepthModel, RegionModality, DepthModality,
// Optimizer, Tracker::StartModalities / CalculateCorrespondences / CalculateGradientAndHessian /
// CalculateOptimization / CalculateResults / ExecuteTrackingStep; M3T/include/m3t/tracker.h:131-160),
// steps return bool and report on std::cerr like the reference.  No Eigen / OpenCV types:
// poses are std::array<float,16> column-major (== Eigen::Transform<float,3,Affine>::data()),
// images are raw pointers + row step (== cv::Mat::data / cv::Mat::step).
// INTEGRATION.md shows the m3t::Modality adapter built with the user's Eigen / OpenCV.
#ifndef M3T_HIP_HPP_
#define M3T_HIP_HPP_

#include <array>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "m3t_hip.h"

namespace m3t_hip {

using Pose = std::array<float, 16>;  // column-major 4x4

inline Pose IdentityPose() { return Pose{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; }

class Context {
 public:
  explicit Context(int device_id = 0) {
    int rc = m3t_hip_create(&ctx_, device_id);
    if (rc != M3T_OK) throw std::runtime_error(std::string("m3t_hip_create: ") + m3t_hip_last_error(nullptr));
  }
  ~Context() { m3t_hip_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  m3t_hip_context* get() const { return ctx_; }
  // constructors throw on invalid arguments (the reference's SetUp() would return false)
  int Check(int rc, const char* what) const {
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + m3t_hip_last_error(ctx_));
    return rc;
  }
  // steps return bool + message on std::cerr, like every Tracker / Modality step of the reference
  bool Step(int rc) const {
    if (rc < 0) {
      std::cerr << m3t_hip_last_error(ctx_) << std::endl;
      return false;
    }
    return true;
  }

 private:
  m3t_hip_context* ctx_ = nullptr;
};
using ContextPtr = std::shared_ptr<Context>;

class Body {
 public:
  Body(ContextPtr c, const Pose& body2world_pose) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_body_create(c_->get(), body2world_pose.data()), "Body");
  }
  void set_body2world_pose(const Pose& p) { c_->Check(m3t_hip_body_set_body2world_pose(c_->get(), id_, p.data()), "Body"); }
  Pose body2world_pose() const {
    Pose p;
    c_->Check(m3t_hip_body_get_body2world_pose(c_->get(), id_, p.data()), "Body");
    return p;
  }
  // the triangle mesh of body.h (vertices in metres), needed only for the renderer-fed branches
  void set_geometry(const m3t_body_geometry& geometry) {
    c_->Check(m3t_hip_body_set_geometry(c_->get(), id_, &geometry), "Body");
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

// Body mesh + renderers for the renderer-fed branches (m3t_hip.h "renderer-fed branches")
class RendererGeometry {
 public:
  explicit RendererGeometry(ContextPtr c) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_renderer_geometry_create(c_->get()), "RendererGeometry");
  }
  bool AddBody(const Body& body) { return c_->Step(m3t_hip_renderer_geometry_add_body(c_->get(), id_, body.id())); }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

class Camera {
 public:
  // Camera::UpdateImage: pixels = BGR8 (color) or u16 (depth), row_step in bytes
  bool UpdateImage(const void* pixels, size_t row_step) {
    return c_->Step(m3t_hip_camera_upload(c_->get(), id_, pixels, row_step));
  }
  void set_world2camera_pose(const Pose& p) {
    c_->Check(m3t_hip_camera_set_world2camera_pose(c_->get(), id_, p.data()), "Camera");
  }
  // device-side frame ring and asynchronous ingest (m3t_hip.h: camera_set_ring ... ingest_sync)
  void SetRing(int n_slots) { c_->Check(m3t_hip_camera_set_ring(c_->get(), id_, n_slots), "Camera"); }
  bool UploadSlot(int slot, const void* pixels, size_t row_step, bool asynchronous = false) {
    return c_->Step(asynchronous ? m3t_hip_camera_upload_slot_async(c_->get(), id_, slot, pixels, row_step)
                                 : m3t_hip_camera_upload_slot(c_->get(), id_, slot, pixels, row_step));
  }
  bool SelectSlot(int slot) { return c_->Step(m3t_hip_camera_select_slot(c_->get(), id_, slot)); }
  bool SlotSync(int slot) { return c_->Step(m3t_hip_camera_slot_sync(c_->get(), id_, slot)); }  // this camera's copy only
  int id() const { return id_; }

 protected:
  ContextPtr c_;
  int id_ = -1;
};
class ColorCamera : public Camera {
 public:
  ColorCamera(ContextPtr c, const m3t_intrinsics& intrinsics, const Pose& world2camera_pose = IdentityPose()) {
    c_ = std::move(c);
    id_ = c_->Check(m3t_hip_color_camera_create(c_->get(), &intrinsics, world2camera_pose.data()), "ColorCamera");
  }
};
class DepthCamera : public Camera {
 public:
  DepthCamera(ContextPtr c, const m3t_intrinsics& intrinsics, float depth_scale,
              const Pose& world2camera_pose = IdentityPose()) {
    c_ = std::move(c);
    id_ = c_->Check(m3t_hip_depth_camera_create(c_->get(), &intrinsics, world2camera_pose.data(), depth_scale),
                    "DepthCamera");
  }
};

// m3t::ColorHistograms shared by several RegionModalities
class ColorHistograms {
 public:
  ColorHistograms(ContextPtr c, int n_bins = 16, float learning_rate_f = 0.2f, float learning_rate_b = 0.2f)
      : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_color_histograms_create(c_->get(), n_bins, learning_rate_f, learning_rate_b),
                    "ColorHistograms");
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

// m3t::FocusedBasicDepthRenderer / m3t::FocusedSilhouetteRenderer (software, no OpenGL context)
class FocusedRenderer {
 public:
  bool AddReferencedBody(const Body& body) {
    return c_->Step(m3t_hip_renderer_add_referenced_body(c_->get(), id_, body.id()));
  }
  bool StartRendering() { return c_->Step(m3t_hip_renderer_start_rendering(c_->get(), id_)); }
  // focused_depth_image() / focused_silhouette_image(): image_size^2 values each (silhouette may be null);
  // info = corner_u, corner_v, scale of the focused crop; returns the number of visible referenced bodies
  int FetchImages(uint16_t* depth, uint8_t* silhouette, float info[3]) const {
    int n_visible = 0;
    c_->Check(m3t_hip_renderer_get_images(c_->get(), id_, depth, silhouette, info, &n_visible), "FocusedRenderer");
    return n_visible;
  }
  int id() const { return id_; }

 protected:
  ContextPtr c_;
  int id_ = -1;
};
class FocusedBasicDepthRenderer : public FocusedRenderer {
 public:
  FocusedBasicDepthRenderer(ContextPtr c, const RendererGeometry& geometry, const Camera& camera, int image_size = 200,
                            float z_min = 0.02f, float z_max = 10.0f) {
    c_ = std::move(c);
    id_ = c_->Check(m3t_hip_focused_depth_renderer_create(c_->get(), geometry.id(), camera.id(), image_size, z_min, z_max),
                    "FocusedBasicDepthRenderer");
  }
};
class FocusedSilhouetteRenderer : public FocusedRenderer {
 public:
  FocusedSilhouetteRenderer(ContextPtr c, const RendererGeometry& geometry, const Camera& camera,
                            int id_type = M3T_ID_TYPE_BODY, int image_size = 200, float z_min = 0.02f,
                            float z_max = 10.0f) {
    c_ = std::move(c);
    id_ = c_->Check(m3t_hip_focused_silhouette_renderer_create(c_->get(), geometry.id(), camera.id(), id_type,
                                                               image_size, z_min, z_max),
                    "FocusedSilhouetteRenderer");
  }
};

class RegionModel {
 public:
  RegionModel(ContextPtr c, const std::string& model_path) : c_(std::move(c)) {  // RegionModel::LoadModel
    id_ = c_->Check(m3t_hip_region_model_load(c_->get(), model_path.c_str()), "RegionModel");
  }
  RegionModel(ContextPtr c, const m3t_region_model_desc& desc) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_region_model_create(c_->get(), &desc), "RegionModel");
  }
  // RegionModel::GenerateModel without OpenGL; the body needs set_geometry()
  RegionModel(ContextPtr c, const Body& body, const m3t_model_generation_params& params) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_region_model_generate(c_->get(), body.id(), &params), "RegionModel");
  }
  // ... with associated bodies (RegionModel::AddAssociatedBody): body ids and their movable / same-region flags
  RegionModel(ContextPtr c, const Body& body, const m3t_model_generation_params& params,
              const std::vector<int>& associated_body_ids, const std::vector<int>& movable,
              const std::vector<int>& same_region)
      : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_region_model_generate_associated(c_->get(), body.id(), &params, int(associated_body_ids.size()),
                                                             associated_body_ids.data(), movable.data(), same_region.data()),
                    "RegionModel");
  }
  int GetClosestView(const Pose& body2camera_pose) const {
    int v = 0;
    c_->Check(m3t_hip_region_model_closest_view(c_->get(), id_, body2camera_pose.data(), &v), "GetClosestView");
    return v;
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};
class DepthModel {
 public:
  DepthModel(ContextPtr c, const std::string& model_path) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_depth_model_load(c_->get(), model_path.c_str()), "DepthModel");
  }
  // DepthModel::GenerateModel without OpenGL; the body needs set_geometry()
  DepthModel(ContextPtr c, const Body& body, const m3t_model_generation_params& params) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_depth_model_generate(c_->get(), body.id(), &params), "DepthModel");
  }
  // ... with occlusion bodies (DepthModel::AddOcclusionBody)
  DepthModel(ContextPtr c, const Body& body, const m3t_model_generation_params& params,
             const std::vector<int>& occlusion_body_ids)
      : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_depth_model_generate_occluded(c_->get(), body.id(), &params, int(occlusion_body_ids.size()),
                                                          occlusion_body_ids.data()),
                    "DepthModel");
  }
  DepthModel(ContextPtr c, const m3t_depth_model_desc& desc) : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_depth_model_create(c_->get(), &desc), "DepthModel");
  }
  int GetClosestView(const Pose& body2camera_pose) const {
    int v = 0;
    c_->Check(m3t_hip_depth_model_closest_view(c_->get(), id_, body2camera_pose.data(), &v), "GetClosestView");
    return v;
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

class Modality {
 public:
  virtual ~Modality() = default;
  // Modality::gradient() / hessian(): 6 floats, column-major 6x6
  void gradient_hessian(float gradient[6], float hessian[36]) const {
    c_->Check(m3t_hip_modality_get_gradient_hessian(c_->get(), id_, gradient, hessian), "Modality");
  }
  // test / adapter hook: feed g/H computed elsewhere into the optimizer
  void set_gradient_hessian(const float gradient[6], const float hessian[36]) {
    c_->Check(m3t_hip_modality_set_gradient_hessian(c_->get(), id_, gradient, hessian), "Modality");
  }
  int id() const { return id_; }

 protected:
  ContextPtr c_;
  int id_ = -1;
};
class RegionModality : public Modality {
 public:
  RegionModality(ContextPtr c, const Body& body, const ColorCamera& color_camera, const RegionModel& region_model,
                 const m3t_region_modality_params& params, const DepthCamera* depth_camera = nullptr) {
    c_ = std::move(c);
    id_ = c_->Check(m3t_hip_region_modality_create(c_->get(), &params, body.id(), color_camera.id(), region_model.id(),
                                                   depth_camera ? depth_camera->id() : -1),
                    "RegionModality");
  }
  bool UseSharedColorHistograms(const ColorHistograms& histograms) {
    return c_->Step(m3t_hip_region_modality_use_shared_color_histograms(c_->get(), id_, histograms.id()));
  }
  bool ModelOcclusions(const FocusedBasicDepthRenderer& renderer) {
    return c_->Step(m3t_hip_region_modality_model_occlusions(c_->get(), id_, renderer.id()));
  }
  bool UseRegionChecking(const FocusedSilhouetteRenderer& renderer) {
    return c_->Step(m3t_hip_region_modality_use_region_checking(c_->get(), id_, renderer.id()));
  }
  // ColorHistograms::histogram_f / histogram_b, n_bins^3 floats each
  void histograms(float* histogram_f, float* histogram_b) const {
    c_->Check(m3t_hip_region_modality_get_histograms(c_->get(), id_, histogram_f, histogram_b), "histograms");
  }
  void set_histograms(const float* histogram_f, const float* histogram_b) {
    c_->Check(m3t_hip_region_modality_set_histograms(c_->get(), id_, histogram_f, histogram_b), "histograms");
  }
  std::vector<m3t_data_line> data_lines(int capacity = 1024) const {
    std::vector<m3t_data_line> out(capacity);
    int n = 0;
    c_->Check(m3t_hip_region_modality_get_lines(c_->get(), id_, out.data(), capacity, &n), "data_lines");
    out.resize(n < capacity ? n : capacity);
    return out;
  }
};
class DepthModality : public Modality {
 public:
  DepthModality(ContextPtr c, const Body& body, const DepthCamera& depth_camera, const DepthModel& depth_model,
                const m3t_depth_modality_params& params) {
    c_ = std::move(c);
    id_ = c_->Check(m3t_hip_depth_modality_create(c_->get(), &params, body.id(), depth_camera.id(), depth_model.id()),
                    "DepthModality");
  }
  bool ModelOcclusions(const FocusedBasicDepthRenderer& renderer) {
    return c_->Step(m3t_hip_depth_modality_model_occlusions(c_->get(), id_, renderer.id()));
  }
  bool UseSilhouetteChecking(const FocusedSilhouetteRenderer& renderer) {
    return c_->Step(m3t_hip_depth_modality_use_silhouette_checking(c_->get(), id_, renderer.id()));
  }
  std::vector<m3t_data_point> data_points(int capacity = 1024) const {
    std::vector<m3t_data_point> out(capacity);
    int n = 0;
    c_->Check(m3t_hip_depth_modality_get_points(c_->get(), id_, out.data(), capacity, &n), "data_points");
    out.resize(n < capacity ? n : capacity);
    return out;
  }
};

// m3t::Optimizer with one free 6-dof root Link holding the modalities of one body
class Optimizer {
 public:
  Optimizer(ContextPtr c, const Body& body, const std::vector<const Modality*>& modalities,
            float tikhonov_parameter_rotation = 1000.0f, float tikhonov_parameter_translation = 30000.0f)
      : c_(std::move(c)) {
    std::vector<int> ids;
    for (auto* m : modalities) ids.push_back(m->id());
    id_ = c_->Check(m3t_hip_optimizer_create_rigid(c_->get(), body.id(), int(ids.size()), ids.data(),
                                                   tikhonov_parameter_rotation, tikhonov_parameter_translation),
                    "Optimizer");
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

// m3t::Link (link.h:67): body and parent are optional (pure joints / root links)
class Link {
 public:
  Link(ContextPtr c, const Body* body = nullptr, const Link* parent = nullptr,
       const Pose& body2joint_pose = IdentityPose(), const Pose& joint2parent_pose = IdentityPose(),
       const std::array<bool, 6>& free_directions = {true, true, true, true, true, true},
       bool fixed_body2joint_pose = true)
      : c_(std::move(c)) {
    int fd[6];
    for (int i = 0; i < 6; ++i) fd[i] = free_directions[i] ? 1 : 0;
    id_ = c_->Check(m3t_hip_link_create(c_->get(), body ? body->id() : -1, parent ? parent->id() : -1,
                                        body2joint_pose.data(), joint2parent_pose.data(), fd,
                                        fixed_body2joint_pose ? 1 : 0),
                    "Link");
  }
  void AddModality(const Modality& m) { c_->Check(m3t_hip_link_add_modality(c_->get(), id_, m.id()), "Link"); }
  void set_joint2parent_pose(const Pose& p) {
    c_->Check(m3t_hip_link_set_joint_poses(c_->get(), id_, nullptr, p.data()), "Link");
  }
  void set_body2joint_pose(const Pose& p) {
    c_->Check(m3t_hip_link_set_joint_poses(c_->get(), id_, p.data(), nullptr), "Link");
  }
  Pose joint2parent_pose() const {
    Pose p;
    c_->Check(m3t_hip_link_get_joint_poses(c_->get(), id_, nullptr, p.data()), "Link");
    return p;
  }
  Pose link2world_pose() const {
    Pose p;
    c_->Check(m3t_hip_link_get_link2world_pose(c_->get(), id_, p.data()), "Link");
    return p;
  }
  void set_link2world_pose(const Pose& p) {
    c_->Check(m3t_hip_link_set_link2world_pose(c_->get(), id_, p.data()), "Link");
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

// m3t::Optimizer over a kinematic tree + m3t::Constraint (constraint.h)
class TreeOptimizer {
 public:
  TreeOptimizer(ContextPtr c, const Link& root_link, float tikhonov_parameter_rotation = 1000.0f,
                float tikhonov_parameter_translation = 30000.0f)
      : c_(std::move(c)) {
    id_ = c_->Check(m3t_hip_optimizer_create(c_->get(), root_link.id(), tikhonov_parameter_rotation,
                                             tikhonov_parameter_translation),
                    "Optimizer");
  }
  int AddConstraint(const Link& link1, const Link& link2, const Pose& body12joint1_pose, const Pose& body22joint2_pose,
                    const std::array<bool, 6>& constraint_directions) {
    int cd[6];
    for (int i = 0; i < 6; ++i) cd[i] = constraint_directions[i] ? 1 : 0;
    return c_->Check(m3t_hip_constraint_create(c_->get(), id_, link1.id(), link2.id(), body12joint1_pose.data(),
                                               body22joint2_pose.data(), cd),
                     "Constraint");
  }
  // m3t::SoftConstraint (soft_constraint.h:52-62, same defaults)
  int AddSoftConstraint(const Link& link1, const Link& link2, const Pose& body12joint1_pose,
                        const Pose& body22joint2_pose, const std::array<bool, 6>& constraint_directions,
                        float max_distance_rotation = 0.0f, float max_distance_translation = 0.0f,
                        float standard_deviation_rotation = 0.01f, float standard_deviation_translation = 0.001f) {
    int cd[6];
    for (int i = 0; i < 6; ++i) cd[i] = constraint_directions[i] ? 1 : 0;
    return c_->Check(m3t_hip_soft_constraint_create(c_->get(), id_, link1.id(), link2.id(), body12joint1_pose.data(),
                                                    body22joint2_pose.data(), cd, max_distance_rotation,
                                                    max_distance_translation, standard_deviation_rotation,
                                                    standard_deviation_translation),
                     "SoftConstraint");
  }
  int id() const { return id_; }

 private:
  ContextPtr c_;
  int id_;
};

// m3t::Tracker restricted to the tracking step (tracker.cpp:344-364, 430-517)
class Tracker {
 public:
  Tracker(ContextPtr c, int n_corr_iterations = 5, int n_update_iterations = 2) : c_(std::move(c)) {
    c_->Check(m3t_hip_tracker_set_iterations(c_->get(), n_corr_iterations, n_update_iterations), "Tracker");
  }
  bool StartModalities(int iteration) { return c_->Step(m3t_hip_start_modalities(c_->get(), iteration)); }
  bool CalculateCorrespondences(int iteration, int corr_iteration) {
    return c_->Step(m3t_hip_calculate_correspondences(c_->get(), iteration, corr_iteration));
  }
  bool CalculateGradientAndHessian(int iteration, int corr_iteration, int update_iteration) {
    return c_->Step(m3t_hip_calculate_gradient_and_hessian(c_->get(), iteration, corr_iteration, update_iteration));
  }
  bool CalculateOptimization(int iteration, int corr_iteration, int update_iteration) {
    return c_->Step(m3t_hip_calculate_optimization(c_->get(), iteration, corr_iteration, update_iteration));
  }
  bool CalculateResults(int iteration) { return c_->Step(m3t_hip_calculate_results(c_->get(), iteration)); }
  bool CalculateConsistentPoses() { return c_->Step(m3t_hip_calculate_consistent_poses(c_->get())); }
  bool ExecuteTrackingStep(int iteration) { return c_->Step(m3t_hip_execute_tracking_step(c_->get(), iteration)); }
  bool ExecuteTrackingCycle(int iteration) { return c_->Step(m3t_hip_execute_tracking_cycle(c_->get(), iteration)); }
  // four workgroups per object for small batches (needs the GPU to itself; results are identical either way)
  void SetObjectSplit(bool enable) { c_->Check(m3t_hip_set_object_split(c_->get(), enable ? 1 : 0), "Tracker"); }
  // m3t::Refiner::RefinePoses (refiner.cpp:76-117)
  bool RefinePoses(int n_corr_iterations = 7, int n_update_iterations = 2) {
    return c_->Step(m3t_hip_refine_poses(c_->get(), n_corr_iterations, n_update_iterations));
  }
  bool Sync() { return c_->Step(m3t_hip_sync(c_->get())); }

  // ---- the batch at once, and what has no counterpart in the reference (m3t_hip.h) ----
  // poses of the first n bodies in creation order, one copy each way
  void SetBodyPoses(const std::vector<Pose>& poses) {
    c_->Check(m3t_hip_bodies_set_poses(c_->get(), poses.empty() ? nullptr : poses[0].data(), int(poses.size())), "Tracker");
  }
  std::vector<Pose> BodyPoses(int n) const {
    std::vector<Pose> poses(static_cast<size_t>(n));
    c_->Check(m3t_hip_bodies_get_poses(c_->get(), n ? poses[0].data() : nullptr, n), "Tracker");
    return poses;
  }
  // frame ring of every camera + asynchronous ingest
  bool SelectSlot(int slot) { return c_->Step(m3t_hip_cameras_select_slot(c_->get(), slot)); }
  void RegisterHostBuffer(void* ptr, size_t bytes) { c_->Check(m3t_hip_host_register(c_->get(), ptr, bytes), "Tracker"); }
  void UnregisterHostBuffer(void* ptr) { c_->Check(m3t_hip_host_unregister(c_->get(), ptr), "Tracker"); }
  bool IngestSync() { return c_->Step(m3t_hip_ingest_sync(c_->get())); }
  // gradient | Hessian of every modality (creation order), one read-back
  std::vector<float> GradientsAndHessians(int n_modalities) const {
    std::vector<float> out(size_t(n_modalities) * 42);
    c_->Check(m3t_hip_modalities_get_gradient_hessian(c_->get(), out.data(), n_modalities), "Tracker");
    return out;
  }
  // batch ingest: one frame ring for a group of cameras, one transfer per batch-frame
  void SetSharedRing(const std::vector<int>& camera_ids, int n_slots) {
    c_->Check(m3t_hip_cameras_set_ring(c_->get(), camera_ids.data(), int(camera_ids.size()), n_slots), "Tracker");
  }
  bool UploadBatchAsync(const std::vector<int>& camera_ids, int slot, const void* base, size_t camera_stride,
                        size_t row_step) {
    return c_->Step(m3t_hip_cameras_upload_batch_async(c_->get(), camera_ids.data(), int(camera_ids.size()), slot, base,
                                                       camera_stride, row_step));
  }
  // ROI ingest: rectangles instead of frames (m3t_hip.h)
  // adaptive: per-body margins from the motion over the last step, at most margin_px
  void SetRoiIngest(bool enable, float margin_px, bool adaptive = false) {
    c_->Check(m3t_hip_set_roi_ingest(c_->get(), enable ? (adaptive ? 2 : 1) : 0, margin_px), "Tracker");
  }
  // ... pulled by a kernel on CUs of its own while the step runs on the others (replaces the streams: stream() again)
  void ReserveIngestCus(int n_cus) { c_->Check(m3t_hip_reserve_ingest_cus(c_->get(), n_cus), "Tracker"); }
  bool UploadBatchRoiAsync(const std::vector<int>& camera_ids, int slot, const void* base, size_t camera_stride,
                           size_t row_step) {
    return c_->Step(m3t_hip_cameras_upload_batch_roi_async(c_->get(), camera_ids.data(), int(camera_ids.size()), slot,
                                                           base, camera_stride, row_step));
  }
  // bodies whose steps left their rectangle since the last call (they were repeated on whole frames: the poses are
  // the whole-frame poses; the list counts how often the margin was too small)
  std::vector<int> RoiMisses() {
    std::vector<int> bodies(256);
    int n = 0;
    c_->Check(m3t_hip_roi_get_status(c_->get(), bodies.data(), int(bodies.size()), &n, nullptr), "Tracker");
    bodies.resize(size_t(n < int(bodies.size()) ? n : int(bodies.size())));
    return bodies;
  }
  std::vector<int> RoiUnrecovered() {  // ... and the ones whose repeat could not see a whole frame either (m3t_hip.h)
    std::vector<int> bodies(256);
    int n = 0;
    c_->Check(m3t_hip_roi_get_unrecovered(c_->get(), bodies.data(), int(bodies.size()), &n), "Tracker");
    bodies.resize(size_t(n < int(bodies.size()) ? n : int(bodies.size())));
    return bodies;
  }
  // a kinematic structure spread over GPUs: begin -> all-reduce(sum) of `count` floats at `partial` on stream() -> end
  bool CalculateOptimizationBegin(float** partial, size_t* count) {
    return c_->Step(m3t_hip_calculate_optimization_begin(c_->get(), partial, count));
  }
  bool CalculateOptimizationEnd() { return c_->Step(m3t_hip_calculate_optimization_end(c_->get())); }
  // ... with the library's own RCCL call site: one rank creates the id, every rank joins; from then on
  // CalculateOptimization / ExecuteTrackingStep sum over the ranks by themselves
  std::vector<char> CommUniqueId() {
    std::vector<char> id(128);
    c_->Check(m3t_hip_comm_get_unique_id(c_->get(), id.data(), id.size()), "Tracker");
    return id;
  }
  void CommInitRank(const std::vector<char>& id, int n_ranks, int rank) {
    c_->Check(m3t_hip_comm_init_rank(c_->get(), id.data(), id.size(), n_ranks, rank), "Tracker");
  }
  void CommSet(void* nccl_comm) { c_->Check(m3t_hip_comm_set(c_->get(), nccl_comm), "Tracker"); }
  void CommDestroy() { c_->Check(m3t_hip_comm_destroy(c_->get()), "Tracker"); }
  // the host's own transport in ncclAllReduce's place (MPI, gloo, threads that play ranks): fn sums the device buffer
  // over the host's ranks, in stream order; nullptr removes it
  void CommSetReduceCallback(m3t_hip_reduce_fn fn, void* user) {
    c_->Check(m3t_hip_comm_set_reduce_callback(c_->get(), fn, user), "Tracker");
  }
  bool CalculateOptimizationAllReduce() { return c_->Step(m3t_hip_calculate_optimization_allreduce(c_->get())); }
  void* stream() const {
    void* s = nullptr;
    c_->Check(m3t_hip_get_stream(c_->get(), &s), "Tracker");
    return s;
  }
  long long AllReduceCount() const {  // ncclAllReduce calls issued so far (one per Newton step with a communicator set)
    long long n = 0;
    c_->Check(m3t_hip_comm_get_allreduce_count(c_->get(), &n), "Tracker");
    return n;
  }
  int CommRankCount() const {  // ncclCommCount of the live communicator, 0 without one
    int n = 0;
    c_->Check(m3t_hip_comm_get_rank_count(c_->get(), &n), "Tracker");
    return n;
  }
  // test hook: checksum and general-logarithm count of the device's logarithm over a range of float bit patterns
  std::array<unsigned long long, 3> DebugLogChecksum(unsigned first_bits, unsigned last_bits) const {
    std::array<unsigned long long, 3> out{};
    c_->Check(m3t_hip_debug_log_checksum(c_->get(), first_bits, last_bits, out.data()), "Tracker");
    return out;
  }
  // launch shape (m3t_hip.h: set_fused_step, get_step_shape)
  void SetFusedStep(int mode) { c_->Check(m3t_hip_set_fused_step(c_->get(), mode), "Tracker"); }
  std::string StepKernel() const {  // name of the kernel the last ExecuteTrackingStep launched for the tracking loop
    char name[64] = {0};
    c_->Check(m3t_hip_get_step_kernel(c_->get(), name, sizeof(name)), "Tracker");
    return name;
  }
  std::array<int, 4> StepShape() const {
    std::array<int, 4> shape{};
    c_->Check(m3t_hip_get_step_shape(c_->get(), shape.data()), "Tracker");
    return shape;
  }
  // measurement aids
  void SetKernelTiming(bool enable) { c_->Check(m3t_hip_set_kernel_timing(c_->get(), enable ? 1 : 0), "Tracker"); }
  void KernelTiming(float total_ms[2], int launches[2]) const {
    c_->Check(m3t_hip_get_kernel_timing(c_->get(), total_ms, launches), "Tracker");
  }
  std::string DeviceName(int* compute_units = nullptr, size_t* total_memory_bytes = nullptr) const {
    char name[256] = {0};
    int cus = 0;
    size_t bytes = 0;
    c_->Check(m3t_hip_device_info(c_->get(), name, sizeof(name), &cus, &bytes), "Tracker");
    if (compute_units) *compute_units = cus;
    if (total_memory_bytes) *total_memory_bytes = bytes;
    return name;
  }

 private:
  ContextPtr c_;
};

}  // namespace m3t_hip
#endif  // M3T_HIP_HPP_
