// Evaluator front-ends of the reference's dataset benchmarks (C++ counterpart of 3dobjecttracking_amd/evaluation.py):
// RBOT pose file reader and 5 cm / 5 degree criterion (examples/rbot_evaluator.cpp:416-433,558-585), YCB-Video pose
// reader, reduced vertices, ADD / ADD-S, loss curves and area under curve (examples/ycb_evaluator.cpp:18-22,803-901,
// 1222-1248).  Header-only, C++17, no dependencies; poses are column-major float[16] like m3t_hip::Pose.
#ifndef M3T_HIP_EVALUATION_HPP_
#define M3T_HIP_EVALUATION_HPP_

#include <array>
#include <cmath>
#include <fstream>
#include <limits>
#include <random>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace m3t_hip {
namespace evaluation {

using Pose = std::array<float, 16>;  // column-major 4x4

// RBOTEvaluator::ReadPosesRBOTDataset: one header line, then per frame nine rotation entries (row-major) and a
// translation in millimetres, tab separated; n_frames + 1 poses
inline std::vector<Pose> ReadPosesRBOT(const std::string& path, int n_frames = 1000) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.is_open()) throw std::runtime_error("Could not open file stream " + path);
  std::string line;
  std::getline(ifs, line);
  std::vector<Pose> poses(size_t(n_frames) + 1);
  for (auto& pose : poses) {
    if (!std::getline(ifs, line)) throw std::runtime_error("Could not read all poses from " + path);
    std::stringstream ss(line);
    std::string item;
    float v[12];
    for (int i = 0; i < 12; ++i) {
      if (!std::getline(ss, item, '\t')) throw std::runtime_error("Could not read all poses from " + path);
      v[i] = std::stof(item);
    }
    pose = Pose{};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) pose[size_t(c * 4 + r)] = v[r * 3 + c];
    for (int r = 0; r < 3; ++r) pose[size_t(12 + r)] = v[9 + r] * 0.001f;
    pose[15] = 1.0f;
  }
  return poses;
}

struct RbotResult {
  float translation_error, rotation_error, tracking_success;
};
// RBOTEvaluator::CalculatePoseResults; thresholds rbot_evaluator.h:192-193
inline RbotResult RbotPoseResult(const Pose& pose, const Pose& gt, float translation_error_threshold = 0.05f,
                                 float rotation_error_threshold = 5.0f * 3.14159265358979323846f / 180.0f) {
  RbotResult r;
  const float dx = pose[12] - gt[12], dy = pose[13] - gt[13], dz = pose[14] - gt[14];
  r.translation_error = std::sqrt(dx * dx + dy * dy + dz * dz);
  float trace = 0.0f;  // trace(R^T R_gt) = sum of the products of corresponding entries
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 3; ++k) trace += pose[size_t(c * 4 + k)] * gt[size_t(c * 4 + k)];
  r.rotation_error = std::acos((trace - 1.0f) / 2.0f);
  r.tracking_success =
      (r.translation_error > translation_error_threshold || r.rotation_error > rotation_error_threshold) ? 0.0f : 1.0f;
  return r;
}

constexpr int kNCurveValues = 100;     // ycb_evaluator.h:45
constexpr float kThresholdMax = 0.1f;  // ycb_evaluator.h:46

inline std::array<float, kNCurveValues> YcbThresholds() {  // ycb_evaluator.cpp:18-22
  std::array<float, kNCurveValues> t{};
  const float step = kThresholdMax / float(kNCurveValues);
  for (int i = 0; i < kNCurveValues; ++i) t[size_t(i)] = step * (0.5f + float(i));
  return t;
}

// YCBEvaluator::LoadGTPoses: 'qw qx qy qz tx ty tz' per frame of every sequence; skip pose_begin lines, keep the
// lines whose 1-based frame index is a keyframe; quaternions normalised, pose = translation * rotation
inline std::vector<Pose> ReadPosesYCB(const std::string& path, int pose_begin, int n_frames,
                                      const std::vector<int>& keyframes) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.is_open()) throw std::runtime_error("Could not open file stream " + path);
  std::string line;
  for (int i = 0; i < pose_begin; ++i) std::getline(ifs, line);
  std::vector<Pose> poses;
  size_t k = 0;
  for (int idx = 1; idx <= n_frames && k < keyframes.size(); ++idx) {
    std::getline(ifs, line);
    if (idx != keyframes[k]) continue;
    std::stringstream ss(line);
    float v[7];
    for (float& x : v) ss >> x;
    const float n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3]);
    const float w = v[0] / n, x = v[1] / n, y = v[2] / n, z = v[3] / n;
    Pose p{};
    p[0] = 1 - 2 * (y * y + z * z); p[4] = 2 * (x * y - z * w);     p[8] = 2 * (x * z + y * w);
    p[1] = 2 * (x * y + z * w);     p[5] = 1 - 2 * (x * x + z * z); p[9] = 2 * (y * z - x * w);
    p[2] = 2 * (x * z - y * w);     p[6] = 2 * (y * z + x * w);     p[10] = 1 - 2 * (x * x + y * y);
    p[12] = v[4]; p[13] = v[5]; p[14] = v[6]; p[15] = 1.0f;
    poses.push_back(p);
    ++k;
  }
  return poses;
}

// YCBEvaluator::GenderateReducedVertices: all vertices, or n draws 'mt19937{7}() % n_vertices' with repetition
inline std::vector<std::array<float, 3>> ReduceVertices(const std::vector<std::array<float, 3>>& vertices,
                                                        int n_vertices_evaluation) {
  if (n_vertices_evaluation <= 0 || size_t(n_vertices_evaluation) >= vertices.size()) return vertices;
  std::mt19937 generator{7};
  std::vector<std::array<float, 3>> reduced(static_cast<size_t>(n_vertices_evaluation));
  const int n = int(vertices.size());
  for (auto& v : reduced) v = vertices[size_t(int(generator() % unsigned(n)))];
  return reduced;
}

struct YcbResult {
  float add_error = 0.0f, adds_error = 0.0f, add_auc = 0.0f, adds_auc = 0.0f;
  std::array<float, kNCurveValues> add_curve{}, adds_curve{};
};
// YCBEvaluator::CalculatePoseResults: delta = body2world^-1 * gt; ADD = mean |v - delta v|, ADD-S = mean distance
// of delta v to the nearest (reduced) vertex — exhaustive search instead of the reference's k-d tree: same minimum
inline YcbResult YcbPoseResult(const std::vector<std::array<float, 3>>& vertices, const Pose& body2world,
                               const Pose& gt_body2world) {
  // rigid inverse in double, then the product, then float like the Transform3fA the reference holds
  double inv[16] = {0};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) inv[c * 4 + r] = body2world[size_t(r * 4 + c)];
  for (int r = 0; r < 3; ++r)
    inv[12 + r] = -(inv[r] * body2world[12] + inv[4 + r] * body2world[13] + inv[8 + r] * body2world[14]);
  inv[15] = 1.0;
  float delta[16];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += inv[k * 4 + r] * double(gt_body2world[size_t(c * 4 + k)]);
      delta[c * 4 + r] = float(s);
    }
  YcbResult out;
  double add = 0.0, adds = 0.0;
  std::vector<std::array<float, 3>> moved(vertices.size());
  for (size_t i = 0; i < vertices.size(); ++i) {
    const auto& v = vertices[i];
    for (int r = 0; r < 3; ++r) moved[i][size_t(r)] = delta[r] * v[0] + delta[4 + r] * v[1] + delta[8 + r] * v[2] + delta[12 + r];
    const float dx = v[0] - moved[i][0], dy = v[1] - moved[i][1], dz = v[2] - moved[i][2];
    add += std::sqrt(dx * dx + dy * dy + dz * dz);
  }
  for (const auto& m : moved) {
    float best = std::numeric_limits<float>::max();
    for (const auto& v : vertices) {
      const float dx = v[0] - m[0], dy = v[1] - m[1], dz = v[2] - m[2];
      best = std::min(best, dx * dx + dy * dy + dz * dz);
    }
    adds += std::sqrt(best);
  }
  out.add_error = float(add / double(vertices.size()));
  out.adds_error = float(adds / double(vertices.size()));
  const auto thresholds = YcbThresholds();
  out.add_curve.fill(1.0f);
  out.adds_curve.fill(1.0f);
  for (int i = 0; i < kNCurveValues && !(out.add_error < thresholds[size_t(i)]); ++i) out.add_curve[size_t(i)] = 0.0f;
  for (int i = 0; i < kNCurveValues && !(out.adds_error < thresholds[size_t(i)]); ++i) out.adds_curve[size_t(i)] = 0.0f;
  out.add_auc = 1.0f - std::min(out.add_error / kThresholdMax, 1.0f);
  out.adds_auc = 1.0f - std::min(out.adds_error / kThresholdMax, 1.0f);
  return out;
}

}  // namespace evaluation
}  // namespace m3t_hip

#endif  // M3T_HIP_EVALUATION_HPP_
