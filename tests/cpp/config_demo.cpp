// Drives include/m3t_hip_config.hpp for tests/test_cpp_config.py.
//   yaml FILE KEY...        print the scalar / matrix at the key path ("Class/0/key" walks sequences by index)
//   obj FILE UNIT           n_vertices n_triangles and the sums of coordinates / indices
//   png FILE                width height channels bytes_per_channel and the sum of all pixel bytes
//   bin FILE REGION OUT     re-write a model file from its parsed contents (byte-identical round trip)
//   rbot POSES N            first / last translation of an RBOT pose file and a criterion check
//   ycb POSES BEGIN N K...  keyframe poses of a YCB pose file
//   adds OBJ N POSE16 GT16  ADD / ADD-S / AUC of a mesh's (reduced) vertices under two poses (row-major 4x4 each)
//   rbot-dataset DATASET EXTERNAL N_FRAMES N_DIVIDES BODY...       EvaluateRbotDataset on sequence a_regular
//   ycb-dataset DATASET EXTERNAL N_DIVIDES N_POINTS N_VERTICES SEQ_ID... -- BODY...   EvaluateYcbDataset
//   track CONFIG            GenerateConfiguredTracker + SetUp + DetectPoses + StartModalities + one step (needs a GPU)
#include <cinttypes>
#include <cstdio>
#include <cstdlib>

#include "m3t_hip_config.hpp"
#include "m3t_hip_datasets.hpp"
#include "m3t_hip_evaluation.hpp"

using namespace m3t_hip;
namespace cfg = m3t_hip::config;

static void PrintNode(const cfg::Node& n) {
  switch (n.kind) {
    case cfg::Node::kScalar: std::printf("%s\n", n.scalar.c_str()); break;
    case cfg::Node::kMatrix:
      std::printf("matrix %d %d", n.rows, n.cols);
      for (double v : n.data) std::printf(" %.9g", v);
      std::printf("\n");
      break;
    case cfg::Node::kSeq:
      std::printf("seq %zu", n.seq.size());
      for (auto& e : n.seq)
        if (e.kind == cfg::Node::kScalar) std::printf(" %s", e.scalar.c_str());
      std::printf("\n");
      break;
    case cfg::Node::kMap:
      std::printf("map %zu", n.map.size());
      for (auto& kv : n.map) std::printf(" %s", kv.first.c_str());
      std::printf("\n");
      break;
    default: std::printf("null\n");
  }
}

int main(int argc, char** argv) {
  try {
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "yaml" && argc >= 3) {
      cfg::Node root = cfg::ReadYaml(argv[2]);
      const cfg::Node* n = &root;
      for (int i = 3; i < argc; ++i) {
        std::string key = argv[i];
        if (n->kind == cfg::Node::kSeq) n = &n->seq.at(size_t(std::atoi(key.c_str())));
        else n = &(*n)[key];
      }
      PrintNode(*n);
      return 0;
    }
    if (mode == "obj" && argc >= 4) {
      cfg::Mesh m = cfg::LoadObj(argv[2], float(std::atof(argv[3])));
      double vs = 0.0;
      long long ts = 0;
      for (float v : m.vertices) vs += v;
      for (int t : m.triangles) ts += t;
      std::printf("%zu %zu %.9g %lld %.9g\n", m.vertices.size() / 3, m.triangles.size() / 3, vs, ts,
                  double(cfg::MaximumBodyDiameter(m, IdentityPose())));
      return 0;
    }
    if (mode == "png" && argc >= 3) {
      cfg::Image im = cfg::DecodePng(argv[2]);
      unsigned long long sum = 0;
      for (uint8_t b : im.pixels) sum += b;
      std::printf("%d %d %d %d %llu\n", im.width, im.height, im.channels, im.bytes_per_channel, sum);
      return 0;
    }
    if (mode == "bin" && argc >= 5) {
      const bool region = std::atoi(argv[3]) != 0;
      std::ifstream ifs(argv[2], std::ios::binary);
      std::string b((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
      size_t off = 5;
      cfg::ModelParameters p;
      int32_t i32 = 0;
      cfg::detail::Get(b, &off, &p.sphere_radius);
      cfg::detail::Get(b, &off, &i32); p.n_divides = i32;
      cfg::detail::Get(b, &off, &i32); p.n_points = i32;
      cfg::detail::Get(b, &off, &p.max_radius_depth_offset);
      cfg::detail::Get(b, &off, &p.stride_depth_offset);
      cfg::detail::Get(b, &off, &p.use_random_seed);
      cfg::detail::Get(b, &off, &i32); p.image_size = i32;
      cfg::BodyData body;
      if (!cfg::detail::GetBody(b, &off, &body)) return 2;
      off += size_t(region ? 5 : 1) * 8;  // no associated bodies in the fixtures
      uint64_t n_views = 0;
      cfg::detail::Get(b, &off, &n_views);
      const size_t pf = size_t(region ? M3T_REGION_POINT_FLOATS : M3T_DEPTH_POINT_FLOATS) * size_t(p.n_points);
      std::vector<float> points(n_views * pf), orientations(n_views * 3), extents(n_views);
      for (uint64_t v = 0; v < n_views; ++v) {
        std::memcpy(&points[v * pf], b.data() + off, pf * 4); off += pf * 4;
        std::memcpy(&orientations[v * 3], b.data() + off, 12); off += 12;
        std::memcpy(&extents[v], b.data() + off, 4); off += 4;
      }
      if (off != b.size()) return 3;
      cfg::WriteModelBin(argv[4], region, p, body, size_t(n_views), points.data(), orientations.data(), extents.data());
      std::printf("%d %d\n", cfg::ModelBinMatches(argv[4], region, p, body) ? 1 : 0,
                  cfg::ModelBinMatches(argv[4], !region, p, body) ? 1 : 0);
      return 0;
    }
    namespace ev = m3t_hip::evaluation;
    if (mode == "rbot" && argc >= 4) {
      auto poses = ev::ReadPosesRBOT(argv[2], std::atoi(argv[3]));
      std::printf("%zu %.9g %.9g %.9g %.9g\n", poses.size(), double(poses[2][12]), double(poses[2][13]),
                  double(poses[2][14]), double(poses[2][4]));
      ev::Pose moved = poses[1];
      moved[14] += 0.049f;
      auto ok = ev::RbotPoseResult(moved, poses[1]);
      moved[14] += 0.002f;
      auto lost = ev::RbotPoseResult(moved, poses[1]);
      auto other = ev::RbotPoseResult(poses[2], poses[1]);  // two unrelated poses: a rotation error far from 0
      std::printf("%.9g %.9g %g %g\n", double(ok.translation_error), double(other.rotation_error),
                  double(ok.tracking_success), double(lost.tracking_success));
      return 0;
    }
    if (mode == "ycb" && argc >= 6) {
      std::vector<int> keyframes;
      for (int i = 5; i < argc; ++i) keyframes.push_back(std::atoi(argv[i]));
      for (auto& p : ev::ReadPosesYCB(argv[2], std::atoi(argv[3]), std::atoi(argv[4]), keyframes)) {
        for (float v : p) std::printf("%.9g ", double(v));
        std::printf("\n");
      }
      return 0;
    }
    if (mode == "adds" && argc >= 36) {
      cfg::Mesh m = cfg::LoadObj(argv[2], 1.0f);
      std::vector<std::array<float, 3>> vertices(m.vertices.size() / 3);
      for (size_t i = 0; i < vertices.size(); ++i) vertices[i] = {m.vertices[3 * i], m.vertices[3 * i + 1], m.vertices[3 * i + 2]};
      vertices = ev::ReduceVertices(vertices, std::atoi(argv[3]));
      ev::Pose a, b;
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
          a[size_t(c * 4 + r)] = float(std::atof(argv[4 + r * 4 + c]));
          b[size_t(c * 4 + r)] = float(std::atof(argv[20 + r * 4 + c]));
        }
      auto r = ev::YcbPoseResult(vertices, a, b);
      int add_zeros = 0, adds_zeros = 0;
      for (float v : r.add_curve) add_zeros += v == 0.0f;
      for (float v : r.adds_curve) adds_zeros += v == 0.0f;
      std::printf("%zu %.9g %.9g %.9g %.9g %d %d %.9g\n", vertices.size(), double(r.add_error), double(r.adds_error),
                  double(r.add_auc), double(r.adds_auc), add_zeros, adds_zeros, double(vertices[0][0]));
      return 0;
    }
    namespace ds = m3t_hip::datasets;
    if (mode == "rbot-dataset" && argc >= 7) {
      cfg::ModelParameters p = ds::RbotModelParameters();
      p.n_divides = std::atoi(argv[5]);
      std::vector<std::string> bodies(argv + 6, argv + argc);
      auto results = ds::EvaluateRbotDataset([] { return std::make_shared<Context>(0); }, argv[2], argv[3], bodies,
                                             {"a_regular"}, std::atoi(argv[4]), ds::RbotRegionParameters(), p);
      for (auto& r : results)
        std::printf("%s %s %.9g %.9g %.9g\n", r.sequence.c_str(), r.body.c_str(), r.tracking_success,
                    r.translation_error, r.rotation_error);
      return 0;
    }
    if (mode == "ycb-dataset" && argc >= 9) {
      cfg::ModelParameters p = ds::YcbModelParameters();
      p.n_divides = std::atoi(argv[4]);
      p.n_points = std::atoi(argv[5]);
      std::vector<int> sequences;
      std::vector<std::string> bodies;
      int i = 7;
      for (; i < argc && std::string(argv[i]) != "--"; ++i) sequences.push_back(std::atoi(argv[i]));
      for (++i; i < argc; ++i) bodies.push_back(argv[i]);
      auto results = ds::EvaluateYcbDataset([] { return std::make_shared<Context>(0); }, argv[2], argv[3], sequences,
                                            bodies, std::atoi(argv[6]), p);
      for (auto& r : results)
        std::printf("%s %s %d %.9g %.9g\n", r.sequence.c_str(), r.body.c_str(), r.n_frames, r.add_auc, r.adds_auc);
      return 0;
    }
    if (mode == "track" && argc >= 3) {
      auto context = std::make_shared<Context>(0);
      auto tracker = cfg::GenerateConfiguredTracker(context, argv[2]);
      std::set<std::string> names;
      for (auto& o : tracker->optimizers) names.insert(o.first);
      if (tracker->DetectPoses(names)) return 4;  // must refuse before SetUp
      if (!tracker->SetUp() || !tracker->DetectPoses(names) || !tracker->StartModalities(0) ||
          !tracker->ExecuteTrackingStep(0))
        return 5;
      for (auto& b : tracker->bodies) {
        Pose p = b.second->body2world_pose();
        std::printf("%s %d %d", b.first.c_str(), tracker->n_corr_iterations, tracker->n_update_iterations);
        for (float v : p) std::printf(" %a", double(v));
        std::printf("\n");
      }
      for (auto& mp : tracker->model_paths) std::printf("model %s %s\n", mp.first.c_str(), mp.second.c_str());
      return 0;
    }
    // RunTrackerProcess-style loop over the loaded sequence (tracker.cpp:209-330): detect, start, then a step per frame
    // until the images run out; `roi` != 0: the loader cameras hand their frames over as rectangles (margin_px = roi)
    if (mode == "process" && argc >= 4) {
      auto context = std::make_shared<Context>(0);
      auto tracker = cfg::GenerateConfiguredTracker(context, argv[2]);
      const float roi = float(std::atof(argv[3]));
      if (roi != 0.0f && !tracker->EnableRoiIngest(true, roi, false)) return 6;
      std::set<std::string> names;
      for (auto& o : tracker->optimizers) names.insert(o.first);
      if (!tracker->SetUp() || !tracker->DetectPoses(names) || !tracker->StartModalities(0)) return 5;
      int steps = 0;
      for (int k = 0; k < 8; ++k) {
        if (k > 0 && !tracker->UpdateCameras()) break;
        if (!tracker->ExecuteTrackingStep(k)) return 7;
        ++steps;
        for (auto& b : tracker->bodies) {
          Pose p = b.second->body2world_pose();
          std::printf("%d %s", k, b.first.c_str());
          for (float v : p) std::printf(" %a", double(v));
          std::printf("\n");
        }
      }
      int n = 0;
      long long pulls = 0;
      std::vector<int> bodies(16);
      m3t_hip_roi_get_status(context->get(), bodies.data(), 16, &n, &pulls);
      std::printf("steps %d rectangle_uploads %lld repeated %d\n", steps, pulls, n);
      return 0;
    }
    std::fprintf(stderr, "usage: config_demo yaml|obj|png|bin|track|process ...\n");
    return 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 10;
  }
}
