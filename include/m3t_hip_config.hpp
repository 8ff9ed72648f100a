// C++ configuration front-end over include/m3t_hip.hpp: what a host needs either side of the tracking path to
// run the reference's YAML configurations without OpenCV / OpenGL.
//
//   ReadYaml                cv::FileStorage's YAML subset (common.cpp:84-100): '%YAML:1.2', block mappings and
//                           sequences, flow sequences / mappings, quoted strings, '!!opencv-matrix' nodes
//   LoadObj                 Body::LoadMeshData (body.cpp:185-242)
//   BodyData, WriteModelBin, ModelBinMatches   sparse viewpoint model files (model.cpp:218-323,
//                           region_model.cpp:259-363, depth_model.cpp:215-291)
//   DecodePng               cv::imread(IMREAD_UNCHANGED) for the 8-bit colour / 16-bit depth PNGs of the datasets
//   LoaderColorCamera, LoaderDepthCamera       loader_camera.cpp
//   StaticDetector          static_detector.cpp, detector.cpp:42-53
//   GenerateConfiguredTracker                  generator.h:943-1133 for the classes of the tracking path
//
// Header-only, C++17; link with -lm3t_hip -lz.  Errors are std::runtime_error with the reference's message text
// where the reference prints a message and returns false.  The Python package holds the same front-end
// (3dobjecttracking_amd/config.py, generator.py); both are checked against the reference's tracker_config.yaml.
#ifndef M3T_HIP_CONFIG_HPP_
#define M3T_HIP_CONFIG_HPP_

#include <zlib.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "m3t_hip.hpp"

namespace m3t_hip {
namespace config {

inline std::string DirName(const std::string& path) {
  size_t p = path.find_last_of('/');
  return p == std::string::npos ? std::string(".") : (p == 0 ? std::string("/") : path.substr(0, p));
}
// lexical normalisation of dir/rel ("." and ".." folded), like std::filesystem::path::lexically_normal
inline std::string RelativeTo(const std::string& metafile_path, const std::string& p) {
  std::string joined = (!p.empty() && p[0] == '/') ? p : DirName(metafile_path) + "/" + p;
  std::vector<std::string> parts;
  std::stringstream ss(joined);
  std::string item;
  const bool absolute = !joined.empty() && joined[0] == '/';
  while (std::getline(ss, item, '/')) {
    if (item.empty() || item == ".") continue;
    if (item == ".." && !parts.empty() && parts.back() != "..") parts.pop_back();
    else parts.push_back(item);
  }
  std::string out = absolute ? "/" : "";
  for (size_t i = 0; i < parts.size(); ++i) out += (i ? "/" : "") + parts[i];
  return out.empty() ? std::string(".") : out;
}

// ---------------------------------------------------------------------------------------------------------
// YAML
// ---------------------------------------------------------------------------------------------------------
struct Node {
  enum Kind { kNull, kScalar, kSeq, kMap, kMatrix } kind = kNull;
  std::string scalar;
  std::vector<Node> seq;
  std::vector<std::pair<std::string, Node>> map;  // in file order
  int rows = 0, cols = 0;
  std::vector<double> data;  // kMatrix, row-major

  bool has(const std::string& key) const {
    for (auto& kv : map)
      if (kv.first == key) return true;
    return false;
  }
  const Node& operator[](const std::string& key) const {
    static const Node null_node;
    for (auto& kv : map)
      if (kv.first == key) return kv.second;
    return null_node;
  }
  bool empty() const { return kind == kNull; }
  double number() const {
    if (kind != kScalar) throw std::runtime_error("yaml: not a number");
    char* end = nullptr;
    double v = std::strtod(scalar.c_str(), &end);
    if (end == scalar.c_str()) throw std::runtime_error("yaml: not a number: " + scalar);
    return v;
  }
  int integer() const { return int(number()); }
  bool boolean() const { return scalar == "true" || scalar == "True" || (scalar != "false" && scalar != "False" && number() != 0.0); }
  const std::string& str() const { return scalar; }
  std::vector<double> numbers() const {
    std::vector<double> v;
    if (kind == kMatrix) return data;
    for (auto& n : seq) v.push_back(n.number());
    return v;
  }
  // Transform3fA of a 4x4 node: read as double, stored as float, column-major like Eigen
  Pose pose() const {
    std::vector<double> v = numbers();
    if (v.size() != 16) throw std::runtime_error("yaml: a pose needs 16 values");
    Pose p;
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) p[c * 4 + r] = float(v[r * 4 + c]);
    return p;
  }
};

namespace detail {
inline std::string Trim(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
inline std::string Unquote(const std::string& s) {
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
    return s.substr(1, s.size() - 2);
  return s;
}
struct Line {
  int indent;
  std::string text;
};
// flow collections: [a, b, [c]] and {k: v}
inline Node ParseFlow(const std::string& s, size_t* pos);
inline void SkipSpace(const std::string& s, size_t* pos) {
  while (*pos < s.size() && std::isspace((unsigned char)s[*pos])) ++*pos;
}
inline Node ParseFlowScalar(const std::string& s, size_t* pos, const char* stops) {
  SkipSpace(s, pos);
  size_t a = *pos;
  if (a < s.size() && (s[a] == '"' || s[a] == '\'')) {
    size_t b = s.find(s[a], a + 1);
    if (b == std::string::npos) throw std::runtime_error("yaml: unterminated string");
    *pos = b + 1;
    Node n;
    n.kind = Node::kScalar;
    n.scalar = s.substr(a + 1, b - a - 1);
    return n;
  }
  while (*pos < s.size() && !std::strchr(stops, s[*pos])) ++*pos;
  Node n;
  n.kind = Node::kScalar;
  n.scalar = Trim(s.substr(a, *pos - a));
  return n;
}
inline Node ParseFlow(const std::string& s, size_t* pos) {
  SkipSpace(s, pos);
  Node n;
  if (*pos < s.size() && s[*pos] == '[') {
    n.kind = Node::kSeq;
    ++*pos;
    for (;;) {
      SkipSpace(s, pos);
      if (*pos >= s.size()) throw std::runtime_error("yaml: unterminated sequence");
      if (s[*pos] == ']') { ++*pos; break; }
      if (s[*pos] == ',') { ++*pos; continue; }
      n.seq.push_back((s[*pos] == '[' || s[*pos] == '{') ? ParseFlow(s, pos) : ParseFlowScalar(s, pos, ",]"));
    }
  } else if (*pos < s.size() && s[*pos] == '{') {
    n.kind = Node::kMap;
    ++*pos;
    for (;;) {
      SkipSpace(s, pos);
      if (*pos >= s.size()) throw std::runtime_error("yaml: unterminated mapping");
      if (s[*pos] == '}') { ++*pos; break; }
      if (s[*pos] == ',') { ++*pos; continue; }
      Node key = ParseFlowScalar(s, pos, ":,}");
      SkipSpace(s, pos);
      if (*pos >= s.size() || s[*pos] != ':') throw std::runtime_error("yaml: ':' expected in flow mapping");
      ++*pos;
      SkipSpace(s, pos);
      Node value = (*pos < s.size() && (s[*pos] == '[' || s[*pos] == '{')) ? ParseFlow(s, pos)
                                                                             : ParseFlowScalar(s, pos, ",}");
      n.map.emplace_back(key.scalar, value);
    }
  } else {
    n = ParseFlowScalar(s, pos, "");
  }
  return n;
}
inline int Balance(const std::string& s) {
  int depth = 0;
  char quote = 0;
  for (char c : s) {
    if (quote) { if (c == quote) quote = 0; continue; }
    if (c == '"' || c == '\'') quote = c;
    else if (c == '[' || c == '{') ++depth;
    else if (c == ']' || c == '}') --depth;
  }
  return depth;
}
inline Node ParseBlock(std::vector<Line>& lines, size_t* idx, int indent);
// the value after "key:" (rest of the line, possibly continued on the following lines)
inline Node ParseValue(std::vector<Line>& lines, size_t* idx, int indent, std::string rest) {
  rest = Trim(rest);
  bool matrix = false;
  if (rest.rfind("!!opencv-matrix", 0) == 0) {
    matrix = true;
    rest = Trim(rest.substr(15));
  }
  Node n;
  if (rest.empty()) {
    if (*idx < lines.size() && lines[*idx].indent > indent) n = ParseBlock(lines, idx, lines[*idx].indent);
  } else if (rest[0] == '[' || rest[0] == '{') {
    while (Balance(rest) > 0 && *idx < lines.size()) rest += " " + lines[(*idx)++].text;
    size_t pos = 0;
    n = ParseFlow(rest, &pos);
  } else {
    n.kind = Node::kScalar;
    n.scalar = Unquote(rest);
  }
  if (matrix) {
    Node m;
    m.kind = Node::kMatrix;
    m.rows = n["rows"].integer();
    m.cols = n["cols"].integer();
    m.data = n["data"].numbers();
    if (int(m.data.size()) != m.rows * m.cols) throw std::runtime_error("yaml: opencv-matrix with wrong data size");
    return m;
  }
  return n;
}
inline size_t KeyEnd(const std::string& t) {  // position of the ':' that ends a block-mapping key
  char quote = 0;
  for (size_t i = 0; i < t.size(); ++i) {
    char c = t[i];
    if (quote) { if (c == quote) quote = 0; continue; }
    if (c == '"' || c == '\'') quote = c;
    else if (c == ':' && (i + 1 == t.size() || t[i + 1] == ' ')) return i;
  }
  return std::string::npos;
}
inline Node ParseBlock(std::vector<Line>& lines, size_t* idx, int indent) {
  Node n;
  if (*idx >= lines.size()) return n;
  if (lines[*idx].text.rfind("- ", 0) == 0 || lines[*idx].text == "-") {
    n.kind = Node::kSeq;
    while (*idx < lines.size() && lines[*idx].indent == indent &&
           (lines[*idx].text.rfind("- ", 0) == 0 || lines[*idx].text == "-")) {
      std::string rest = Trim(lines[*idx].text.substr(1));
      if (!rest.empty() && rest[0] != '[' && rest[0] != '{' && KeyEnd(rest) != std::string::npos) {
        // "- key: value": the item is a mapping that starts on this line, two columns further in
        lines[*idx].indent = indent + 2;
        lines[*idx].text = rest;
        n.seq.push_back(ParseBlock(lines, idx, indent + 2));
      } else {
        ++*idx;
        n.seq.push_back(ParseValue(lines, idx, indent, rest));
      }
    }
    return n;
  }
  n.kind = Node::kMap;
  while (*idx < lines.size() && lines[*idx].indent == indent) {
    const std::string t = lines[*idx].text;
    size_t colon = KeyEnd(t);
    if (colon == std::string::npos) throw std::runtime_error("yaml: 'key: value' expected in line: " + t);
    std::string key = Unquote(Trim(t.substr(0, colon)));
    ++*idx;
    n.map.emplace_back(key, ParseValue(lines, idx, indent, t.substr(colon + 1)));
  }
  return n;
}
}  // namespace detail

// OpenYamlFileStorage (common.cpp:84-100)
inline Node ReadYaml(const std::string& path) {
  std::ifstream ifs(path);
  if (!ifs.is_open()) throw std::runtime_error("Could not open file " + path);
  std::vector<detail::Line> lines;
  std::string raw;
  while (std::getline(ifs, raw)) {
    if (raw.rfind("%YAML", 0) == 0 || raw.rfind("---", 0) == 0) continue;
    // strip comments outside quotes
    char quote = 0;
    for (size_t i = 0; i < raw.size(); ++i) {
      if (quote) { if (raw[i] == quote) quote = 0; continue; }
      if (raw[i] == '"' || raw[i] == '\'') quote = raw[i];
      else if (raw[i] == '#' && (i == 0 || raw[i - 1] == ' ')) { raw.resize(i); break; }
    }
    std::string t = detail::Trim(raw);
    if (t.empty()) continue;
    lines.push_back({int(raw.find_first_not_of(' ')), t});
  }
  size_t idx = 0;
  if (lines.empty()) return Node{};
  Node root = detail::ParseBlock(lines, &idx, lines[0].indent);
  if (idx != lines.size()) throw std::runtime_error("Could not parse " + path + " near: " + lines[idx].text);
  return root;
}

inline void Required(const Node& n, std::initializer_list<const char*> keys, const std::string& what,
                     const std::string& path) {
  for (const char* k : keys)
    if (!n.has(k))
      throw std::runtime_error("Could not read all required " + what + " parameters from " + path + " (missing " + k +
                               ")");
}

// ---------------------------------------------------------------------------------------------------------
// meshes
// ---------------------------------------------------------------------------------------------------------
struct Mesh {
  std::vector<float> vertices;  // x y z, metres
  std::vector<int> triangles;   // three vertex indices each, in the file's winding
};
// Body::LoadMeshData (body.cpp:185-242): 'v' and 'f' records, polygons split as a fan
inline Mesh LoadObj(const std::string& path, float geometry_unit_in_meter = 1.0f) {
  std::ifstream ifs(path);
  if (!ifs.is_open()) throw std::runtime_error("TinyObjLoader failed to load data from " + path);
  Mesh m;
  std::string line;
  while (std::getline(ifs, line)) {
    std::stringstream ss(line);
    std::string tag;
    ss >> tag;
    if (tag == "v") {
      float x, y, z;
      ss >> x >> y >> z;
      m.vertices.insert(m.vertices.end(), {x, y, z});
    } else if (tag == "f") {
      std::vector<int> idx;
      std::string corner;
      while (ss >> corner) {
        int i = std::atoi(corner.substr(0, corner.find('/')).c_str());
        idx.push_back(i > 0 ? i - 1 : int(m.vertices.size() / 3) + i);
      }
      for (size_t k = 1; k + 1 < idx.size(); ++k) m.triangles.insert(m.triangles.end(), {idx[0], idx[k], idx[k + 1]});
    }
  }
  if (m.vertices.empty() || m.triangles.empty()) throw std::runtime_error("TinyObjLoader failed to load data from " + path);
  if (geometry_unit_in_meter != 1.0f)
    for (auto& v : m.vertices) v *= geometry_unit_in_meter;
  return m;
}
// Body::CalculateMaximumBodyDiameter (body.cpp:244-252)
inline float MaximumBodyDiameter(const Mesh& m, const Pose& g2b) {
  float max_radius = 0.0f;
  for (size_t i = 0; i + 2 < m.vertices.size(); i += 3) {
    const float x = m.vertices[i], y = m.vertices[i + 1], z = m.vertices[i + 2];
    const float px = g2b[0] * x + g2b[4] * y + g2b[8] * z + g2b[12];
    const float py = g2b[1] * x + g2b[5] * y + g2b[9] * z + g2b[13];
    const float pz = g2b[2] * x + g2b[6] * y + g2b[10] * z + g2b[14];
    max_radius = std::max(max_radius, std::sqrt(px * px + py * py + pz * pz));
  }
  return 2.0f * max_radius;
}

// ---------------------------------------------------------------------------------------------------------
// sparse viewpoint model files
// ---------------------------------------------------------------------------------------------------------
struct ModelParameters {  // model.h:132-138
  float sphere_radius = 0.8f;
  int n_divides = 4;
  int n_points = 200;
  float max_radius_depth_offset = 0.05f;
  float stride_depth_offset = 0.002f;
  bool use_random_seed = false;
  int image_size = 2000;
};
struct BodyData {  // model.cpp:301-323
  std::string geometry_path;
  float geometry_unit_in_meter = 1.0f;
  bool geometry_counterclockwise = true, geometry_enable_culling = true;
  float maximum_body_diameter = 0.0f;
  Pose geometry2body_pose = IdentityPose();
};
namespace detail {
template <typename T>
inline void Put(std::string* out, const T& v) { out->append(reinterpret_cast<const char*>(&v), sizeof(T)); }
template <typename T>
inline bool Get(const std::string& b, size_t* off, T* v) {
  if (*off + sizeof(T) > b.size()) return false;
  std::memcpy(v, b.data() + *off, sizeof(T));
  *off += sizeof(T);
  return true;
}
inline void PutBody(std::string* out, const BodyData& d) {
  Put(out, uint64_t(d.geometry_path.size()));
  out->append(d.geometry_path);
  Put(out, d.geometry_unit_in_meter);
  Put(out, d.geometry_counterclockwise);
  Put(out, d.geometry_enable_culling);
  Put(out, d.maximum_body_diameter);
  out->append(reinterpret_cast<const char*>(d.geometry2body_pose.data()), 64);
}
inline bool GetBody(const std::string& b, size_t* off, BodyData* d) {
  uint64_t n = 0;
  if (!Get(b, off, &n) || *off + n > b.size()) return false;
  d->geometry_path = b.substr(*off, n);
  *off += n;
  if (!Get(b, off, &d->geometry_unit_in_meter) || !Get(b, off, &d->geometry_counterclockwise) ||
      !Get(b, off, &d->geometry_enable_culling) || !Get(b, off, &d->maximum_body_diameter))
    return false;
  if (*off + 64 > b.size()) return false;
  std::memcpy(d->geometry2body_pose.data(), b.data() + *off, 64);
  *off += 64;
  return true;
}
inline void PutParameters(std::string* out, bool region, const ModelParameters& p) {
  Put(out, char(region ? 'r' : 'd'));
  Put(out, int32_t(region ? 10 : 9));  // kVersionID region_model.h / depth_model.h
  Put(out, p.sphere_radius);
  Put(out, int32_t(p.n_divides));
  Put(out, int32_t(p.n_points));
  Put(out, p.max_radius_depth_offset);
  Put(out, p.stride_depth_offset);
  Put(out, p.use_random_seed);
  Put(out, int32_t(p.image_size));
}
inline bool SameFile(const std::string& a, const std::string& b) {  // common.cpp Equivalent()
  if (a == b) return true;
  std::ifstream fa(a, std::ios::binary), fb(b, std::ios::binary);
  return fa.is_open() && fb.is_open() && RelativeTo("/", a) == RelativeTo("/", b);
}
}  // namespace detail
inline bool SameBody(const BodyData& a, const BodyData& b) {
  return detail::SameFile(a.geometry_path, b.geometry_path) && a.geometry_unit_in_meter == b.geometry_unit_in_meter &&
         a.geometry_counterclockwise == b.geometry_counterclockwise &&
         a.geometry_enable_culling == b.geometry_enable_culling && a.maximum_body_diameter == b.maximum_body_diameter &&
         a.geometry2body_pose == b.geometry2body_pose;
}
// RegionModel / DepthModel::SaveModel for a model without associated / occlusion bodies
inline void WriteModelBin(const std::string& path, bool region, const ModelParameters& p, const BodyData& body,
                          size_t n_views, const float* points, const float* orientations, const float* extents) {
  std::string head;
  detail::PutParameters(&head, region, p);
  detail::PutBody(&head, body);
  for (int i = 0; i < (region ? 5 : 1); ++i) detail::Put(&head, uint64_t(0));  // no associated bodies
  detail::Put(&head, uint64_t(n_views));
  std::error_code ec;
  std::filesystem::create_directories(DirName(path), ec);  // (the reference expects the directory to exist)
  std::ofstream ofs(path, std::ios::binary);
  if (!ofs.is_open()) throw std::runtime_error("Could not open model file " + path);
  ofs.write(head.data(), std::streamsize(head.size()));
  const size_t point_floats = size_t(region ? M3T_REGION_POINT_FLOATS : M3T_DEPTH_POINT_FLOATS) * size_t(p.n_points);
  for (size_t v = 0; v < n_views; ++v) {
    ofs.write(reinterpret_cast<const char*>(points + v * point_floats), std::streamsize(point_floats * 4));
    ofs.write(reinterpret_cast<const char*>(orientations + v * 3), 12);
    ofs.write(reinterpret_cast<const char*>(extents + v), 4);
  }
}
// the acceptance test of Model::LoadModelParameters / LoadBodyData (model.cpp:218-284)
inline bool ModelBinMatches(const std::string& path, bool region, const ModelParameters& p, const BodyData& body) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.is_open()) return false;
  std::string b(1 << 16, '\0');
  ifs.read(&b[0], std::streamsize(b.size()));
  b.resize(size_t(ifs.gcount()));
  std::string want;
  detail::PutParameters(&want, region, p);
  if (b.compare(0, want.size(), want) != 0) return false;
  size_t off = want.size();
  BodyData have;
  if (!detail::GetBody(b, &off, &have) || !SameBody(have, body)) return false;
  for (int i = 0; i < (region ? 5 : 1); ++i) {
    uint64_t n = 1;
    if (!detail::Get(b, &off, &n) || n != 0) return false;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------------------
// PNG (non-interlaced; grey 8 / 16, RGB 8, RGBA 8, palette 8)
// ---------------------------------------------------------------------------------------------------------
struct Image {
  int width = 0, height = 0, channels = 0, bytes_per_channel = 0;
  std::vector<uint8_t> pixels;  // colour: B, G, R per pixel (cv::imread order); 16-bit: host byte order
  size_t row_step() const { return size_t(width) * channels * bytes_per_channel; }
};
inline Image DecodePng(const std::string& path) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.is_open()) throw std::runtime_error("Could not read image from " + path);
  std::string f((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
  static const unsigned char kSig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (f.size() < 8 || std::memcmp(f.data(), kSig, 8) != 0) throw std::runtime_error("Could not read image from " + path);
  auto be32 = [&](size_t o) {
    return (uint32_t(uint8_t(f[o])) << 24) | (uint32_t(uint8_t(f[o + 1])) << 16) | (uint32_t(uint8_t(f[o + 2])) << 8) |
           uint32_t(uint8_t(f[o + 3]));
  };
  int width = 0, height = 0, depth = 0, color = 0, interlace = 0;
  std::string idat, palette;
  for (size_t o = 8; o + 12 <= f.size();) {
    uint32_t len = be32(o);
    std::string type = f.substr(o + 4, 4);
    if (o + 12 + len > f.size()) break;
    if (type == "IHDR") {
      width = int(be32(o + 8));
      height = int(be32(o + 12));
      depth = uint8_t(f[o + 16]);
      color = uint8_t(f[o + 17]);
      interlace = uint8_t(f[o + 20]);
    } else if (type == "PLTE") {
      palette = f.substr(o + 8, len);
    } else if (type == "IDAT") {
      idat.append(f, o + 8, len);
    } else if (type == "IEND") {
      break;
    }
    o += 12 + len;
  }
  const int src_channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
  if (!width || !height || !src_channels || interlace || (depth != 8 && !(depth == 16 && color == 0)))
    throw std::runtime_error("Could not read image from " + path + " (unsupported PNG layout)");
  const size_t bpp = size_t(src_channels) * depth / 8, stride = size_t(width) * bpp;
  std::vector<uint8_t> raw((stride + 1) * size_t(height));
  uLongf raw_len = uLongf(raw.size());
  if (uncompress(raw.data(), &raw_len, reinterpret_cast<const Bytef*>(idat.data()), uLong(idat.size())) != Z_OK ||
      raw_len != raw.size())
    throw std::runtime_error("Could not read image from " + path + " (corrupt data)");
  std::vector<uint8_t> img(stride * size_t(height));
  for (int y = 0; y < height; ++y) {  // undo the scanline filters
    const uint8_t type = raw[(stride + 1) * y];
    const uint8_t* in = &raw[(stride + 1) * y + 1];
    uint8_t* out = &img[stride * y];
    const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
    for (size_t x = 0; x < stride; ++x) {
      const int a = x >= bpp ? out[x - bpp] : 0, b = up ? up[x] : 0, c = (up && x >= bpp) ? up[x - bpp] : 0;
      int pred = 0;
      switch (type) {
        case 0: pred = 0; break;
        case 1: pred = a; break;
        case 2: pred = b; break;
        case 3: pred = (a + b) / 2; break;
        case 4: {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: throw std::runtime_error("Could not read image from " + path + " (bad filter)");
      }
      out[x] = uint8_t(in[x] + pred);
    }
  }
  Image im;
  im.width = width;
  im.height = height;
  if (color == 0) {  // grey: depth images
    im.channels = 1;
    im.bytes_per_channel = depth / 8;
    im.pixels = img;
    if (depth == 16) {
      uint16_t* p = reinterpret_cast<uint16_t*>(im.pixels.data());
      for (size_t i = 0; i < size_t(width) * height; ++i) p[i] = uint16_t((img[2 * i] << 8) | img[2 * i + 1]);
    }
  } else {  // colour: B, G, R (alpha dropped)
    if (color == 4) throw std::runtime_error("Could not read image from " + path + " (grey + alpha)");
    im.channels = 3;
    im.bytes_per_channel = 1;
    im.pixels.resize(size_t(width) * height * 3);
    for (size_t i = 0; i < size_t(width) * height; ++i) {
      const uint8_t* s = color == 3 ? reinterpret_cast<const uint8_t*>(palette.data()) + 3 * img[i] : &img[i * bpp];
      if (color == 3 && 3 * size_t(img[i]) + 2 >= palette.size()) throw std::runtime_error("bad palette index in " + path);
      im.pixels[3 * i] = s[2];
      im.pixels[3 * i + 1] = s[1];
      im.pixels[3 * i + 2] = s[0];
    }
  }
  return im;
}

// ---------------------------------------------------------------------------------------------------------
// objects with metafiles
// ---------------------------------------------------------------------------------------------------------
inline Pose InversePose(const Pose& p) {  // Transform3fA::inverse(), Affine: general 3x3 inverse, in double
  double m[9], inv[9];
  for (int c = 0; c < 3; ++c)
    for (int r = 0; r < 3; ++r) m[r * 3 + c] = p[c * 4 + r];
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                     m[2] * (m[3] * m[7] - m[4] * m[6]);
  inv[0] = (m[4] * m[8] - m[5] * m[7]) / det; inv[1] = (m[2] * m[7] - m[1] * m[8]) / det; inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
  inv[3] = (m[5] * m[6] - m[3] * m[8]) / det; inv[4] = (m[0] * m[8] - m[2] * m[6]) / det; inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
  inv[6] = (m[3] * m[7] - m[4] * m[6]) / det; inv[7] = (m[1] * m[6] - m[0] * m[7]) / det; inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
  Pose out = IdentityPose();
  for (int r = 0; r < 3; ++r) {
    for (int c = 0; c < 3; ++c) out[c * 4 + r] = float(inv[r * 3 + c]);
    out[12 + r] = float(-(inv[r * 3] * p[12] + inv[r * 3 + 1] * p[13] + inv[r * 3 + 2] * p[14]));
  }
  return out;
}

struct LoaderSettings {  // loader_camera.h
  std::string load_directory, image_name_pre, image_name_post, load_image_type = "png";
  int load_index = 0, n_leading_zeros = 0;
  // <pre><zero-padded index><post>.<type> (loader_camera.cpp:83-88)
  std::string ImagePath() const {
    std::string s = std::to_string(load_index);
    int n_zeros = std::max(n_leading_zeros - int(s.size()), 0);
    return load_directory + "/" + image_name_pre + std::string(size_t(n_zeros), '0') + s + image_name_post + "." +
           load_image_type;
  }
};
namespace detail {
inline m3t_intrinsics Intrinsics(const Node& i, const std::string& path) {
  Required(i, {"f_u", "f_v", "pp_x", "pp_y", "width", "height"}, "intrinsics", path);
  m3t_intrinsics out{};
  out.fu = float(i["f_u"].number());
  out.fv = float(i["f_v"].number());
  out.ppu = float(i["pp_x"].number());
  out.ppv = float(i["pp_y"].number());
  out.width = i["width"].integer();
  out.height = i["height"].integer();
  return out;
}
inline LoaderSettings Loader(const Node& d, const std::string& path) {
  LoaderSettings s;
  s.load_directory = RelativeTo(path, d["load_directory"].str());
  if (d.has("image_name_pre")) s.image_name_pre = d["image_name_pre"].str();
  if (d.has("image_name_post")) s.image_name_post = d["image_name_post"].str();
  if (d.has("load_image_type")) s.load_image_type = d["load_image_type"].str();
  if (d.has("load_index")) s.load_index = d["load_index"].integer();
  if (d.has("n_leading_zeros")) s.n_leading_zeros = d["n_leading_zeros"].integer();
  return s;
}
}  // namespace detail

// Overlapped ingest for loader cameras (replaces the blocking imread + upload of
// loader_camera.cpp:76-98).  Sequences are read in order: while frame k is tracked, a worker thread decodes frame
// k + 2 into a page-locked slab and frame k + 1 crosses PCIe on the library's copy stream into the next slot of a
// three-slot device ring; UpdateImage(k) is then a pointer switch.  The worker never touches the C-ABI (a context
// belongs to one host thread).  Where the asynchronous entry points do not exist (a host linked against the CPU
// restatement) the camera falls back to the blocking path: same pixels, same order, same results.
class FramePipeline {
 public:
  static constexpr int kSlots = 3;
  using Check = std::function<std::string(const Image&)>;  // "" or why the image does not fit the camera
  ~FramePipeline() { Drop(nullptr); }
  bool enabled = true;
  // ROI ingest (m3t_hip.h: m3t_hip_set_roi_ingest is a setting of the whole context -- give every loader camera of a
  // tracker the same one): of frame k + 1 only the rectangle the trackers can read is pulled out of the slab while
  // frame k is tracked; a body that outruns its rectangle is repeated on the whole frame by the library, so the poses
  // are those of whole frames bit for bit.  reserve_cus (a multiple of 32): CUs kept free for the pull kernel.
  bool roi = false;
  bool EnableRoi(m3t_hip_context* ctx, bool enable, float margin_px = 24.0f, bool adaptive = false, int reserve_cus = 0) {
    Drop(ctx);
    if (m3t_hip_set_roi_ingest(ctx, enable ? (adaptive ? 2 : 1) : 0, margin_px) < 0) return false;
    if ((reserve_cus || !enable) && m3t_hip_reserve_ingest_cus(ctx, enable ? reserve_cus : 0) < 0) return false;
    roi = enable;
    return true;
  }
  // returns false (and leaves `enabled` false) when the library has no asynchronous ingest
  bool Prepare(m3t_hip_context* ctx, int camera_id, size_t frame_bytes) {
    if (!enabled || ready_) return enabled;
    for (auto& slab : slabs_) slab.assign(frame_bytes, 0);
    // (a ring of the batch kind with this camera as its only member: what the rectangle upload needs)
    if (m3t_hip_cameras_set_ring(ctx, &camera_id, 1, kSlots) < 0) { enabled = false; return false; }
    for (auto& slab : slabs_)
      if (m3t_hip_host_register(ctx, slab.data(), slab.size()) < 0) { enabled = false; return false; }
    ready_ = true;
    return true;
  }
  void Drop(m3t_hip_context* ctx) {
    for (auto& j : jobs_) if (j.second.thread.joinable()) j.second.thread.join();
    jobs_.clear();
    if (ctx && !uploaded_.empty()) (void)m3t_hip_ingest_sync(ctx);
    uploaded_.clear();
  }
  // the owner goes away before its context: the slabs must not stay page-locked behind freed memory
  void Release(m3t_hip_context* ctx) {
    Drop(ctx);
    if (ready_ && ctx)
      for (auto& slab : slabs_) (void)m3t_hip_host_unregister(ctx, slab.data());
    ready_ = false;
  }
  void StartDecode(int index, const std::string& path, const Check& check) {
    if (jobs_.count(index) || uploaded_.count(index)) return;
    Job& job = jobs_[index];
    job.path = path;
    std::vector<uint8_t>* slab = &slabs_[size_t(index % kSlots)];
    Job* j = &job;  // (std::map nodes do not move)
    job.thread = std::thread([j, slab, path, check]() {
      try {
        Image image = DecodePng(path);
        j->error = check(image);
        if (j->error.empty() && image.pixels.size() == slab->size()) {
          std::memcpy(slab->data(), image.pixels.data(), slab->size());
          j->width = image.width; j->height = image.height; j->channels = image.channels;
          j->bytes_per_channel = image.bytes_per_channel;
          j->ok = true;
        } else if (j->error.empty()) {
          j->error = "Could not read image from " + path + " (unexpected size)";
        }
      } catch (const std::exception& e) {
        j->error = e.what();
      }
      j->finished.store(true);
    });
  }
  // main thread: hand a decoded slab to the copy stream; false if it is not there (yet) or could not be decoded
  bool UploadWhenDecoded(m3t_hip_context* ctx, int camera_id, int index, bool wait, size_t row_step, std::string* error) {
    if (uploaded_.count(index)) return true;
    auto it = jobs_.find(index);
    if (it == jobs_.end()) return false;
    if (!wait && !it->second.done()) return false;
    it->second.thread.join();
    const bool ok = it->second.ok;
    if (!ok && error) *error = it->second.error;
    jobs_.erase(it);
    if (!ok) return false;
    const int slot = index % kSlots;
    const int rc = roi ? m3t_hip_cameras_upload_batch_roi_async(ctx, &camera_id, 1, slot, slabs_[size_t(slot)].data(),
                                                                slabs_[size_t(slot)].size(), row_step)
                       : m3t_hip_camera_upload_slot_async(ctx, camera_id, slot, slabs_[size_t(slot)].data(), row_step);
    if (rc < 0) {
      if (error) *error = m3t_hip_last_error(ctx);
      return false;
    }
    uploaded_[index] = slot;
    return true;
  }
  int TakeUploaded(int index) {
    const int slot = uploaded_.at(index);
    uploaded_.erase(index);
    return slot;
  }
  const std::vector<uint8_t>& slab(int slot) const { return slabs_[size_t(slot)]; }
  // One UpdateImage of a loader camera through the pipeline: frame `settings.load_index` becomes the camera's
  // current frame (and `image`), the next two are on their way.  false + message when the frame cannot be read.
  bool Update(m3t_hip_context* ctx, int camera_id, LoaderSettings* settings, const Check& check, size_t row_step,
              Image* image, int width, int height, int channels, int bytes_per_channel) {
    if (settings->load_index != expected_) Drop(ctx);  // the index was set from outside: start over
    const int k = settings->load_index;
    auto path_of = [&](int index) {
      LoaderSettings t = *settings;
      t.load_index = index;
      return t.ImagePath();
    };
    StartDecode(k, path_of(k), check);
    std::string error;
    if (!UploadWhenDecoded(ctx, camera_id, k, true, row_step, &error)) {
      std::cerr << error << std::endl;
      Drop(ctx);
      expected_ = -1;
      return false;
    }
    const int slot = TakeUploaded(k);
    // every copy issued so far has left its slab: the slab frame k + 2 is decoded into (frame k - 1's) is free
    // (a slab a rectangle was pulled from is read again if a body outruns it: also wait for the step that read it)
    if (m3t_hip_ingest_sync(ctx) < 0 || (roi && m3t_hip_camera_slot_sync(ctx, camera_id, (k + 2) % kSlots) < 0) ||
        m3t_hip_camera_select_slot(ctx, camera_id, slot) < 0) {
      std::cerr << m3t_hip_last_error(ctx) << std::endl;
      return false;
    }
    image->width = width; image->height = height; image->channels = channels; image->bytes_per_channel = bytes_per_channel;
    image->pixels = slab(slot);
    settings->load_index++;
    expected_ = settings->load_index;
    StartDecode(k + 1, path_of(k + 1), check);
    (void)UploadWhenDecoded(ctx, camera_id, k + 1, false, row_step, nullptr);
    StartDecode(k + 2, path_of(k + 2), check);
    return true;
  }

 private:
  int expected_ = -1;
  struct Job {
    std::thread thread;
    std::string path, error;
    bool ok = false;
    int width = 0, height = 0, channels = 0, bytes_per_channel = 0;
    bool done() const { return finished.load(); }
    std::atomic<bool> finished{false};
  };
  std::map<int, Job> jobs_;
  std::map<int, int> uploaded_;
  std::array<std::vector<uint8_t>, kSlots> slabs_;
  bool ready_ = false;
};

class LoaderColorCamera : public ColorCamera {
 public:
  LoaderColorCamera(ContextPtr c, const LoaderSettings& loader_settings, const m3t_intrinsics& intrinsics,
                    const Pose& camera2world_pose = IdentityPose())
      : ColorCamera(std::move(c), intrinsics, InversePose(camera2world_pose)),
        settings(loader_settings),
        intrinsics_(intrinsics) {}
  static std::shared_ptr<LoaderColorCamera> FromMetafile(ContextPtr c, const std::string& path) {
    Node d = ReadYaml(path);
    Required(d, {"load_directory", "intrinsics"}, "body", path);
    return std::make_shared<LoaderColorCamera>(std::move(c), detail::Loader(d, path),
                                               detail::Intrinsics(d["intrinsics"], path),
                                               d.has("camera2world_pose") ? d["camera2world_pose"].pose() : IdentityPose());
  }
  FramePipeline pipeline;  // pipeline.enabled = false: the blocking reference behaviour
  ~LoaderColorCamera() { pipeline.Release(c_->get()); }
  bool UpdateImage() {  // LoaderColorCamera::UpdateImage loader_camera.cpp:76-98
    const size_t frame_bytes = size_t(intrinsics_.width) * size_t(intrinsics_.height) * 3;
    if (pipeline.enabled && pipeline.Prepare(c_->get(), id_, frame_bytes)) {
      const m3t_intrinsics in = intrinsics_;
      return pipeline.Update(c_->get(), id_, &settings, [in](const Image& im) {
        return (im.channels != 3 || im.width != in.width || im.height != in.height)
                   ? std::string("Could not read image (not a colour image of the camera's size)") : std::string();
      }, size_t(intrinsics_.width) * 3, &image, intrinsics_.width, intrinsics_.height, 3, 1);
    }
    const std::string path = settings.ImagePath();
    try {
      image = DecodePng(path);
    } catch (const std::exception& e) {
      std::cerr << e.what() << std::endl;
      return false;
    }
    if (image.channels != 3 || image.width != intrinsics_.width || image.height != intrinsics_.height) {
      std::cerr << "Could not read image from " << path << " (not a colour image of the camera's size)" << std::endl;
      return false;
    }
    settings.load_index++;
    return Camera::UpdateImage(image.pixels.data(), image.row_step());
  }
  LoaderSettings settings;
  Image image;
  std::string name;

 private:
  m3t_intrinsics intrinsics_;
};

class LoaderDepthCamera : public DepthCamera {
 public:
  LoaderDepthCamera(ContextPtr c, const LoaderSettings& loader_settings, const m3t_intrinsics& intrinsics,
                    float depth_scale, const Pose& camera2world_pose = IdentityPose())
      : DepthCamera(std::move(c), intrinsics, depth_scale, InversePose(camera2world_pose)),
        settings(loader_settings),
        intrinsics_(intrinsics) {}
  static std::shared_ptr<LoaderDepthCamera> FromMetafile(ContextPtr c, const std::string& path) {
    Node d = ReadYaml(path);
    Required(d, {"load_directory", "intrinsics", "depth_scale"}, "body", path);
    return std::make_shared<LoaderDepthCamera>(std::move(c), detail::Loader(d, path),
                                               detail::Intrinsics(d["intrinsics"], path), float(d["depth_scale"].number()),
                                               d.has("camera2world_pose") ? d["camera2world_pose"].pose() : IdentityPose());
  }
  FramePipeline pipeline;
  ~LoaderDepthCamera() { pipeline.Release(c_->get()); }
  bool UpdateImage() {
    const size_t frame_bytes = size_t(intrinsics_.width) * size_t(intrinsics_.height) * 2;
    if (pipeline.enabled && pipeline.Prepare(c_->get(), id_, frame_bytes)) {
      const m3t_intrinsics in = intrinsics_;
      return pipeline.Update(c_->get(), id_, &settings, [in](const Image& im) {
        return (im.channels != 1 || im.bytes_per_channel != 2 || im.width != in.width || im.height != in.height)
                   ? std::string("Could not read image (not a 16-bit depth image of the camera's size)") : std::string();
      }, size_t(intrinsics_.width) * 2, &image, intrinsics_.width, intrinsics_.height, 1, 2);
    }
    const std::string path = settings.ImagePath();
    try {
      image = DecodePng(path);
    } catch (const std::exception& e) {
      std::cerr << e.what() << std::endl;
      return false;
    }
    if (image.channels != 1 || image.bytes_per_channel != 2 || image.width != intrinsics_.width ||
        image.height != intrinsics_.height) {
      std::cerr << "Could not read image from " << path << " (not a 16-bit depth image of the camera's size)" << std::endl;
      return false;
    }
    settings.load_index++;
    return Camera::UpdateImage(image.pixels.data(), image.row_step());
  }
  LoaderSettings settings;
  Image image;
  std::string name;

 private:
  m3t_intrinsics intrinsics_;
};

// m3t::Body with its mesh (body.cpp:13-42,152-252)
class MeshBody : public Body {
 public:
  MeshBody(ContextPtr c, const std::string& body_name, const BodyData& body_data, int id_body, int id_region)
      : Body(std::move(c), IdentityPose()), name(body_name), data(body_data), body_id(id_body), region_id(id_region) {
    mesh = LoadObj(data.geometry_path, data.geometry_unit_in_meter);
    data.maximum_body_diameter = MaximumBodyDiameter(mesh, data.geometry2body_pose);
    m3t_body_geometry g{};
    g.vertices = mesh.vertices.data();
    g.n_vertices = int(mesh.vertices.size() / 3);
    g.triangles = mesh.triangles.data();
    g.n_triangles = int(mesh.triangles.size() / 3);
    std::memcpy(g.geometry2body, data.geometry2body_pose.data(), 64);
    g.geometry_counterclockwise = data.geometry_counterclockwise ? 1 : 0;
    g.geometry_enable_culling = data.geometry_enable_culling ? 1 : 0;
    g.body_id = body_id;
    g.region_id = region_id;
    set_geometry(g);
  }
  static std::shared_ptr<MeshBody> FromMetafile(ContextPtr c, const std::string& name, const std::string& path,
                                                int default_id) {
    Node d = ReadYaml(path);
    Required(d, {"geometry_path", "geometry_unit_in_meter", "geometry_counterclockwise", "geometry_enable_culling",
                 "geometry2body_pose"},
             "body", path);
    BodyData data;
    data.geometry_path = d["geometry_path"].str() == "INFER_FROM_NAME" ? DirName(path) + "/" + name + ".obj"
                                                                         : RelativeTo(path, d["geometry_path"].str());
    data.geometry_unit_in_meter = float(d["geometry_unit_in_meter"].number());
    data.geometry_counterclockwise = d["geometry_counterclockwise"].boolean();
    data.geometry_enable_culling = d["geometry_enable_culling"].boolean();
    data.geometry2body_pose = d["geometry2body_pose"].pose();
    const int body_id = d.has("body_id") ? d["body_id"].integer() : default_id;
    const int region_id = d.has("region_id") ? d["region_id"].integer() : body_id;
    return std::make_shared<MeshBody>(std::move(c), name, data, body_id, region_id);
  }
  std::string name;
  BodyData data;
  Mesh mesh;
  int body_id, region_id;
};

namespace detail {
inline ModelParameters ModelMeta(const Node& d) {
  ModelParameters p;
  if (d.has("sphere_radius")) p.sphere_radius = float(d["sphere_radius"].number());
  if (d.has("n_divides")) p.n_divides = d["n_divides"].integer();
  if (d.has("n_points")) p.n_points = d["n_points"].integer();
  if (d.has("max_radius_depth_offset")) p.max_radius_depth_offset = float(d["max_radius_depth_offset"].number());
  if (d.has("stride_depth_offset")) p.stride_depth_offset = float(d["stride_depth_offset"].number());
  if (d.has("use_random_seed")) p.use_random_seed = d["use_random_seed"].boolean();
  if (d.has("image_size")) p.image_size = d["image_size"].integer();
  return p;
}
inline m3t_model_generation_params Generation(const ModelParameters& p) {
  m3t_model_generation_params g;
  m3t_model_generation_params_default(&g);
  g.sphere_radius = p.sphere_radius;
  g.n_divides = p.n_divides;
  g.n_points = p.n_points;
  g.max_radius_depth_offset = p.max_radius_depth_offset;
  g.stride_depth_offset = p.stride_depth_offset;
  g.image_size = p.image_size;
  return g;
}
}  // namespace detail

// Model::SetUp (region_model.cpp:28-56, depth_model.cpp:28-56): load model_path if it was generated with these
// parameters for this body, else generate on the device and save.  MODEL = RegionModel or DepthModel.
template <typename MODEL, bool REGION>
std::shared_ptr<MODEL> ModelFromMetafile(ContextPtr c, const std::string& name, const std::string& path,
                                         const MeshBody& body, std::string* model_path_out = nullptr) {
  Node d = ReadYaml(path);
  Required(d, {"model_path"}, "body", path);
  const ModelParameters p = detail::ModelMeta(d);
  if (p.use_random_seed) throw std::runtime_error("use_random_seed: models are generated with the fixed seed only");
  const std::string model_path = d["model_path"].str() == "INFER_FROM_NAME" ? DirName(path) + "/" + name + ".bin"
                                                                             : RelativeTo(path, d["model_path"].str());
  if (model_path_out) *model_path_out = model_path;
  if (ModelBinMatches(model_path, REGION, p, body.data)) return std::make_shared<MODEL>(c, model_path);
  auto model = std::make_shared<MODEL>(c, static_cast<const Body&>(body), detail::Generation(p));
  int n_views = 0, n_points = 0;
  float extent = 0.0f;
  c->Check(REGION ? m3t_hip_region_model_info(c->get(), model->id(), &n_views, &n_points, &extent)
                  : m3t_hip_depth_model_info(c->get(), model->id(), &n_views, &n_points, &extent),
           "Model");
  const size_t floats = size_t(REGION ? M3T_REGION_POINT_FLOATS : M3T_DEPTH_POINT_FLOATS);
  std::vector<float> points(size_t(n_views) * n_points * floats), orientations(size_t(n_views) * 3), extents(n_views);
  c->Check(REGION ? m3t_hip_region_model_get_views(c->get(), model->id(), points.data(), orientations.data(), extents.data())
                  : m3t_hip_depth_model_get_views(c->get(), model->id(), points.data(), orientations.data(), extents.data()),
           "Model");
  try {
    WriteModelBin(model_path, REGION, p, body.data, size_t(n_views), points.data(), orientations.data(), extents.data());
  } catch (const std::exception& e) {  // a read-only data directory does not stop tracking
    std::cerr << e.what() << std::endl;
  }
  return model;
}

// ---------------------------------------------------------------------------------------------------------
// the generated tracker
// ---------------------------------------------------------------------------------------------------------
struct ConfiguredLink {  // m3t::Link with the joint poses it was created with (Link::ResetJointPoses link.cpp:243-246)
  std::string name, parent;
  std::shared_ptr<Link> link;
  Pose default_body2joint_pose = IdentityPose(), default_joint2parent_pose = IdentityPose();
  std::vector<std::string> children;
};
struct ConfiguredOptimizer {
  std::string name, root_link_name;
  std::shared_ptr<TreeOptimizer> optimizer;
  std::shared_ptr<Link> root_link;
};
struct StaticDetector {  // static_detector.cpp + Detector::UpdatePoses detector.cpp:42-53
  std::string name, optimizer_name, root_link_name;
  std::shared_ptr<Link> root_link;
  Pose link2world_pose = IdentityPose();
  bool reset_joint_poses = true;
};
struct ConfiguredRefiner {  // refiner.h
  std::string name;
  int n_corr_iterations = 7, n_update_iterations = 2;
};

class GeneratedTracker {
 public:
  std::string name;
  ContextPtr context;
  std::unique_ptr<Tracker> tracker;
  std::map<std::string, std::shared_ptr<MeshBody>> bodies;
  std::map<std::string, std::shared_ptr<LoaderColorCamera>> color_cameras;
  std::map<std::string, std::shared_ptr<LoaderDepthCamera>> depth_cameras;
  std::map<std::string, std::shared_ptr<RegionModel>> region_models;
  std::map<std::string, std::shared_ptr<DepthModel>> depth_models;
  std::map<std::string, std::string> model_paths;
  std::map<std::string, std::shared_ptr<ColorHistograms>> color_histograms;
  std::map<std::string, std::shared_ptr<RendererGeometry>> renderer_geometries;
  std::map<std::string, std::shared_ptr<FocusedBasicDepthRenderer>> depth_renderers;
  std::map<std::string, std::shared_ptr<FocusedSilhouetteRenderer>> silhouette_renderers;
  std::map<std::string, std::shared_ptr<Modality>> modalities;
  std::map<std::string, ConfiguredLink> links;
  std::map<std::string, ConfiguredOptimizer> optimizers;
  std::vector<StaticDetector> detectors;
  std::vector<ConfiguredRefiner> refiners;
  std::vector<std::string> ignored;  // viewers
  int n_corr_iterations = 5, n_update_iterations = 2;

  bool SetUp() {  // the cameras load their first image (Camera::SetUp)
    for (auto& c : color_cameras)
      if (!c.second->UpdateImage()) return false;
    for (auto& c : depth_cameras)
      if (!c.second->UpdateImage()) return false;
    set_up_ = true;
    return true;
  }
  // ROI ingest for every loader camera (FramePipeline::EnableRoi): of each frame only the trackers' rectangle crosses
  // PCIe; poses equal those of whole frames bit for bit (a body that outruns its rectangle is repeated on the whole
  // frame inside ExecuteTrackingStep)
  bool EnableRoiIngest(bool enable, float margin_px = 24.0f, bool adaptive = false, int reserve_cus = 0) {
    bool ok = true;
    for (auto& c : color_cameras) ok = c.second->pipeline.EnableRoi(context->get(), enable, margin_px, adaptive, reserve_cus) && ok;
    for (auto& c : depth_cameras) ok = c.second->pipeline.EnableRoi(context->get(), enable, margin_px, adaptive, reserve_cus) && ok;
    return ok;
  }
  bool UpdateCameras() {
    if (!CheckSetUp()) return false;
    bool ok = true;
    for (auto& c : color_cameras) ok = c.second->UpdateImage() && ok;
    for (auto& c : depth_cameras) ok = c.second->UpdateImage() && ok;
    return ok;
  }
  bool DetectPoses(const std::set<std::string>& names) {
    if (!CheckSetUp()) return false;
    for (auto& d : detectors) {
      if (!names.count(d.optimizer_name)) continue;
      d.root_link->set_link2world_pose(d.link2world_pose);
      if (d.reset_joint_poses) ResetJointPoses(d.root_link_name);
    }
    return tracker->CalculateConsistentPoses();
  }
  // m3t::Refiner::RefinePoses with the first configured refiner's iteration counts (refiner.cpp:76-117)
  bool RefinePoses() {
    const ConfiguredRefiner r = refiners.empty() ? ConfiguredRefiner{} : refiners[0];
    return CheckSetUp() && tracker->RefinePoses(r.n_corr_iterations, r.n_update_iterations);
  }
  bool StartModalities(int iteration) { return CheckSetUp() && tracker->StartModalities(iteration); }
  bool ExecuteTrackingStep(int iteration) { return CheckSetUp() && tracker->ExecuteTrackingStep(iteration); }

 private:
  void ResetJointPoses(const std::string& link_name) {  // the links an optimizer references: the root's subtree
    ConfiguredLink& l = links.at(link_name);
    l.link->set_body2joint_pose(l.default_body2joint_pose);
    l.link->set_joint2parent_pose(l.default_joint2parent_pose);
    for (auto& child : l.children) ResetJointPoses(child);
  }
  bool CheckSetUp() const {
    if (!set_up_) std::cerr << "Set up tracker " << name << " first" << std::endl;
    return set_up_;
  }
  bool set_up_ = false;
};

namespace detail {
inline std::vector<const Node*> Entries(const Node& root, const std::string& class_name,
                                        std::initializer_list<const char*> required, const std::string& path) {
  std::vector<const Node*> out;
  for (auto& n : root[class_name].seq) {
    for (const char* k : required)
      if (!n.has(k))
        throw std::runtime_error(std::string("Required parameter \"") + k + "\" was not found for class " + class_name +
                                 " in " + path);
    out.push_back(&n);
  }
  return out;
}
template <typename MAP>
auto Find(const MAP& m, const std::string& name, const std::string& by) -> decltype(m.begin()->second) {
  auto it = m.find(name);
  if (it == m.end()) throw std::runtime_error("Object " + name + " required by " + by + " was not found");
  return it->second;
}
inline void SetFloats(const Node& n, float* out, int* count, int capacity) {
  std::vector<double> v = n.numbers();
  *count = int(std::min<size_t>(v.size(), size_t(capacity)));
  for (int i = 0; i < *count; ++i) out[i] = float(v[i]);
}
inline void SetInts(const Node& n, int* out, int* count, int capacity) {
  std::vector<double> v = n.numbers();
  *count = int(std::min<size_t>(v.size(), size_t(capacity)));
  for (int i = 0; i < *count; ++i) out[i] = int(v[i]);
}
#define M3T_CFG_F(field) if (m.has(#field)) p.field = float(m[#field].number())
#define M3T_CFG_I(field) if (m.has(#field)) p.field = m[#field].integer()
#define M3T_CFG_B(field) if (m.has(#field)) p.field = m[#field].boolean() ? 1 : 0
inline void RegionMeta(const Node& m, m3t_region_modality_params& p) {  // region_modality.cpp:810-850
  M3T_CFG_I(n_lines_max); M3T_CFG_B(use_adaptive_coverage); M3T_CFG_F(reference_contour_length);
  M3T_CFG_F(min_continuous_distance); M3T_CFG_I(function_length); M3T_CFG_I(distribution_length);
  M3T_CFG_F(function_amplitude); M3T_CFG_F(function_slope); M3T_CFG_F(learning_rate); M3T_CFG_I(n_global_iterations);
  if (m.has("scales")) SetInts(m["scales"], p.scales, &p.n_scales, M3T_MAX_SCALES);
  if (m.has("standard_deviations"))
    SetFloats(m["standard_deviations"], p.standard_deviations, &p.n_standard_deviations, M3T_MAX_SCALES);
  M3T_CFG_I(n_histogram_bins); M3T_CFG_F(learning_rate_f); M3T_CFG_F(learning_rate_b);
  M3T_CFG_F(unconsidered_line_length); M3T_CFG_F(max_considered_line_length); M3T_CFG_F(measured_depth_offset_radius);
  M3T_CFG_F(measured_occlusion_radius); M3T_CFG_F(measured_occlusion_threshold); M3T_CFG_F(modeled_depth_offset_radius);
  M3T_CFG_F(modeled_occlusion_radius); M3T_CFG_F(modeled_occlusion_threshold); M3T_CFG_I(n_unoccluded_iterations);
  M3T_CFG_I(min_n_unoccluded_lines);
}
inline void DepthMeta(const Node& m, m3t_depth_modality_params& p) {  // depth_modality.cpp:560-600
  M3T_CFG_I(n_points_max); M3T_CFG_B(use_adaptive_coverage); M3T_CFG_B(use_depth_scaling);
  M3T_CFG_F(reference_surface_area); M3T_CFG_F(stride_length);
  if (m.has("considered_distances"))
    SetFloats(m["considered_distances"], p.considered_distances, &p.n_considered_distances, M3T_MAX_SCALES);
  if (m.has("standard_deviations"))
    SetFloats(m["standard_deviations"], p.standard_deviations, &p.n_standard_deviations, M3T_MAX_SCALES);
  M3T_CFG_B(measure_occlusions); M3T_CFG_F(measured_depth_offset_radius); M3T_CFG_F(measured_occlusion_radius);
  M3T_CFG_F(measured_occlusion_threshold); M3T_CFG_F(modeled_depth_offset_radius); M3T_CFG_F(modeled_occlusion_radius);
  M3T_CFG_F(modeled_occlusion_threshold); M3T_CFG_I(n_unoccluded_iterations); M3T_CFG_I(min_n_unoccluded_points);
}
#undef M3T_CFG_F
#undef M3T_CFG_I
#undef M3T_CFG_B
}  // namespace detail

// generator.h:943-1133 for the classes of the tracking path: Body, ColorHistograms, RendererGeometry, loader
// cameras, focused depth / silhouette renderers, Region / Depth models and modalities with their measured and
// modelled occlusion options, region / silhouette checking and shared histograms, Link trees, Constraint,
// SoftConstraint, Optimizer, StaticDetector, Refiner, Tracker; viewers are ignored, texture modality, manual
// detector and sensor cameras refused by name.  Objects are created in the order of the Python generator
// (3dobjecttracking_amd/generator.py), so both front-ends hand out the same device ids.
inline std::unique_ptr<GeneratedTracker> GenerateConfiguredTracker(ContextPtr c, const std::string& path) {
  const Node root = ReadYaml(path);
  for (const char* k : {"TextureModality", "ManualDetector", "RealSenseColorCamera", "RealSenseDepthCamera",
                        "AzureKinectColorCamera", "AzureKinectDepthCamera"})
    if (!root[k].seq.empty())
      throw std::runtime_error(std::string("Class ") + k + " of " + path +
                               " is outside the tracking path this library replaces");
  auto t = std::make_unique<GeneratedTracker>();
  t->context = c;
  auto meta = [&](const Node& n) { return RelativeTo(path, n["metafile_path"].str()); };
  int next_id = 1;  // body.cpp:11
  for (auto* n : detail::Entries(root, "Body", {"name", "metafile_path"}, path))
    t->bodies[(*n)["name"].str()] = MeshBody::FromMetafile(c, (*n)["name"].str(), meta(*n), next_id++);
  for (auto* n : detail::Entries(root, "ColorHistograms", {"name"}, path)) {
    Node m = n->has("metafile_path") ? ReadYaml(meta(*n)) : Node{};
    t->color_histograms[(*n)["name"].str()] = std::make_shared<ColorHistograms>(
        c, m.has("n_bins") ? m["n_bins"].integer() : 16, m.has("learning_rate_f") ? float(m["learning_rate_f"].number()) : 0.2f,
        m.has("learning_rate_b") ? float(m["learning_rate_b"].number()) : 0.2f);
  }
  for (auto* n : detail::Entries(root, "RendererGeometry", {"name", "bodies"}, path)) {
    auto geometry = std::make_shared<RendererGeometry>(c);
    for (auto& b : (*n)["bodies"].seq)
      geometry->AddBody(*detail::Find(t->bodies, b.str(), "RendererGeometry " + (*n)["name"].str()));
    t->renderer_geometries[(*n)["name"].str()] = geometry;
  }
  for (auto* n : detail::Entries(root, "LoaderColorCamera", {"name", "metafile_path"}, path))
    (t->color_cameras[(*n)["name"].str()] = LoaderColorCamera::FromMetafile(c, meta(*n)))->name = (*n)["name"].str();
  for (auto* n : detail::Entries(root, "LoaderDepthCamera", {"name", "metafile_path"}, path))
    (t->depth_cameras[(*n)["name"].str()] = LoaderDepthCamera::FromMetafile(c, meta(*n)))->name = (*n)["name"].str();
  auto camera_of = [&](const std::string& name, const std::string& by) -> const Camera& {
    auto ci = t->color_cameras.find(name);
    if (ci != t->color_cameras.end()) return *ci->second;
    return *detail::Find(t->depth_cameras, name, by);
  };
  for (int silhouette = 0; silhouette < 2; ++silhouette) {
    const std::string class_name = silhouette ? "FocusedSilhouetteRenderer" : "FocusedBasicDepthRenderer";
    for (auto* n : detail::Entries(root, class_name, {"name", "renderer_geometry", "camera", "referenced_bodies"}, path)) {
      const std::string name = (*n)["name"].str(), by = class_name + " " + name;
      Node m = n->has("metafile_path") ? ReadYaml(meta(*n)) : Node{};
      const auto& geometry = *detail::Find(t->renderer_geometries, (*n)["renderer_geometry"].str(), by);
      const Camera& camera = camera_of((*n)["camera"].str(), by);
      const int image_size = m.has("image_size") ? m["image_size"].integer() : 200;
      const float z_min = m.has("z_min") ? float(m["z_min"].number()) : 0.02f;
      const float z_max = m.has("z_max") ? float(m["z_max"].number()) : 10.0f;
      FocusedRenderer* renderer;
      if (silhouette) {
        auto r = std::make_shared<FocusedSilhouetteRenderer>(c, geometry, camera, m.has("id_type") ? m["id_type"].integer() : 0,
                                                             image_size, z_min, z_max);
        t->silhouette_renderers[name] = r;
        renderer = r.get();
      } else {
        auto r = std::make_shared<FocusedBasicDepthRenderer>(c, geometry, camera, image_size, z_min, z_max);
        t->depth_renderers[name] = r;
        renderer = r.get();
      }
      for (auto& b : (*n)["referenced_bodies"].seq) renderer->AddReferencedBody(*detail::Find(t->bodies, b.str(), by));
    }
  }
  for (auto* n : detail::Entries(root, "RegionModel", {"name", "metafile_path", "body"}, path)) {
    const std::string name = (*n)["name"].str();
    for (const char* k : {"fixed_bodies", "movable_bodies", "fixed_same_region_bodies", "movable_same_region_bodies"})
      if (n->has(k)) throw std::runtime_error("RegionModel " + name + ": associated bodies (" + k + ") are not generated");
    t->region_models[name] = ModelFromMetafile<RegionModel, true>(
        c, name, meta(*n), *detail::Find(t->bodies, (*n)["body"].str(), "RegionModel " + name), &t->model_paths[name]);
  }
  for (auto* n : detail::Entries(root, "DepthModel", {"name", "metafile_path", "body"}, path)) {
    const std::string name = (*n)["name"].str();
    if (n->has("occlusion_bodies")) throw std::runtime_error("DepthModel " + name + ": occlusion bodies are not generated");
    t->depth_models[name] = ModelFromMetafile<DepthModel, false>(
        c, name, meta(*n), *detail::Find(t->bodies, (*n)["body"].str(), "DepthModel " + name), &t->model_paths[name]);
  }
  std::vector<std::string> region_then_depth;  // (creation order, for readers of the ids)
  for (auto* n : detail::Entries(root, "RegionModality", {"name", "body", "color_camera", "region_model"}, path)) {
    const std::string name = (*n)["name"].str(), by = "RegionModality " + name;
    m3t_region_modality_params p;
    m3t_region_modality_params_default(&p);
    if (n->has("metafile_path")) detail::RegionMeta(ReadYaml(meta(*n)), p);
    std::shared_ptr<LoaderDepthCamera> depth_camera;
    if (n->has("measure_occlusions")) {
      depth_camera = detail::Find(t->depth_cameras, (*n)["measure_occlusions"]["depth_camera"].str(), by);
      p.measure_occlusions = 1;
    }
    auto modality = std::make_shared<RegionModality>(
        c, *detail::Find(t->bodies, (*n)["body"].str(), by), *detail::Find(t->color_cameras, (*n)["color_camera"].str(), by),
        *detail::Find(t->region_models, (*n)["region_model"].str(), by), p, depth_camera.get());
    if (n->has("model_occlusions"))  // (the reference also accepts a silhouette renderer here: it has a depth image)
      modality->ModelOcclusions(*detail::Find(t->depth_renderers, (*n)["model_occlusions"]["focused_depth_renderer"].str(), by));
    if (n->has("use_region_checking"))
      modality->UseRegionChecking(
          *detail::Find(t->silhouette_renderers, (*n)["use_region_checking"]["focused_silhouette_renderer"].str(), by));
    if (n->has("use_shared_color_histograms"))
      modality->UseSharedColorHistograms(
          *detail::Find(t->color_histograms, (*n)["use_shared_color_histograms"]["color_histograms"].str(), by));
    t->modalities[name] = modality;
    region_then_depth.push_back(name);
  }
  for (auto* n : detail::Entries(root, "DepthModality", {"name", "body", "depth_camera", "depth_model"}, path)) {
    const std::string name = (*n)["name"].str(), by = "DepthModality " + name;
    m3t_depth_modality_params p;
    m3t_depth_modality_params_default(&p);
    if (n->has("metafile_path")) detail::DepthMeta(ReadYaml(meta(*n)), p);
    auto modality = std::make_shared<DepthModality>(
        c, *detail::Find(t->bodies, (*n)["body"].str(), by), *detail::Find(t->depth_cameras, (*n)["depth_camera"].str(), by),
        *detail::Find(t->depth_models, (*n)["depth_model"].str(), by), p);
    if (n->has("model_occlusions"))
      modality->ModelOcclusions(*detail::Find(t->depth_renderers, (*n)["model_occlusions"]["focused_depth_renderer"].str(), by));
    if (n->has("use_silhouette_checking"))
      modality->UseSilhouetteChecking(*detail::Find(
          t->silhouette_renderers, (*n)["use_silhouette_checking"]["focused_silhouette_renderer"].str(), by));
    t->modalities[name] = modality;
    region_then_depth.push_back(name);
  }
  // links: parents before children (generator.h:587-641 wires child_links in a second pass)
  std::map<std::string, const Node*> link_nodes;
  std::vector<std::string> link_order;
  std::map<std::string, std::string> parent_of;
  for (auto* n : detail::Entries(root, "Link", {"name"}, path)) {
    link_nodes[(*n)["name"].str()] = n;
    link_order.push_back((*n)["name"].str());
  }
  for (auto& kv : link_nodes)
    for (auto& child : (*kv.second)["child_links"].seq) {
      if (!link_nodes.count(child.str()))
        throw std::runtime_error("Object " + child.str() + " required by Link " + kv.first + " was not found");
      parent_of[child.str()] = kv.first;
    }
  std::function<void(const std::string&, int)> build_link = [&](const std::string& name, int depth) {
    if (t->links.count(name)) return;
    if (depth > int(link_nodes.size())) throw std::runtime_error("Link " + name + " is its own ancestor");
    const Node* n = link_nodes.at(name);
    const std::string by = "Link " + name;
    ConfiguredLink l;
    l.name = name;
    const Link* parent = nullptr;
    if (parent_of.count(name)) {
      build_link(parent_of[name], depth + 1);
      l.parent = parent_of[name];
      parent = t->links.at(l.parent).link.get();
    }
    std::shared_ptr<MeshBody> body;
    if (n->has("body")) body = detail::Find(t->bodies, (*n)["body"].str(), by);
    Node m = n->has("metafile_path") ? ReadYaml(meta(*n)) : Node{};
    std::array<bool, 6> free_directions{true, true, true, true, true, true};
    if (m.has("free_directions")) {
      std::vector<double> v = m["free_directions"].numbers();
      for (size_t i = 0; i < 6 && i < v.size(); ++i) free_directions[i] = v[i] != 0.0;
    }
    if (m.has("body2joint_pose")) l.default_body2joint_pose = m["body2joint_pose"].pose();
    if (m.has("joint2parent_pose")) l.default_joint2parent_pose = m["joint2parent_pose"].pose();
    l.link = std::make_shared<Link>(c, body.get(), parent, l.default_body2joint_pose, l.default_joint2parent_pose,
                                    free_directions,
                                    m.has("fixed_body2joint_pose") ? m["fixed_body2joint_pose"].boolean() : true);
    if (m.has("link2world_pose")) l.link->set_link2world_pose(m["link2world_pose"].pose());
    for (auto& mod : (*n)["modalities"].seq) l.link->AddModality(*detail::Find(t->modalities, mod.str(), by));
    if (!l.parent.empty()) t->links.at(l.parent).children.push_back(name);
    t->links[name] = l;
  };
  for (auto& name : link_order) build_link(name, 0);
  std::map<std::string, const Node*> constraint_nodes, soft_constraint_nodes;
  for (auto* n : detail::Entries(root, "Constraint", {"name", "link1", "link2"}, path)) constraint_nodes[(*n)["name"].str()] = n;
  for (auto* n : detail::Entries(root, "SoftConstraint", {"name", "link1", "link2"}, path))
    soft_constraint_nodes[(*n)["name"].str()] = n;
  auto directions = [](const Node& m) {
    std::array<bool, 6> d{false, false, false, false, false, false};  // constraint.h:111
    if (m.has("constraint_directions")) {
      std::vector<double> v = m["constraint_directions"].numbers();
      for (size_t i = 0; i < 6 && i < v.size(); ++i) d[i] = v[i] != 0.0;
    }
    return d;
  };
  for (auto* n : detail::Entries(root, "Optimizer", {"name", "root_link"}, path)) {
    const std::string name = (*n)["name"].str(), by = "Optimizer " + name;
    Node m = n->has("metafile_path") ? ReadYaml(meta(*n)) : Node{};
    ConfiguredOptimizer o;
    o.name = name;
    o.root_link_name = (*n)["root_link"].str();
    o.root_link = detail::Find(t->links, o.root_link_name, by).link;
    o.optimizer = std::make_shared<TreeOptimizer>(
        c, *o.root_link, m.has("tikhonov_parameter_rotation") ? float(m["tikhonov_parameter_rotation"].number()) : 1000.0f,
        m.has("tikhonov_parameter_translation") ? float(m["tikhonov_parameter_translation"].number()) : 30000.0f);
    for (auto& cn : (*n)["constraints"].seq) {
      const Node& k = *detail::Find(constraint_nodes, cn.str(), by);
      Node km = k.has("metafile_path") ? ReadYaml(meta(k)) : Node{};
      o.optimizer->AddConstraint(*detail::Find(t->links, k["link1"].str(), "Constraint " + cn.str()).link,
                                 *detail::Find(t->links, k["link2"].str(), "Constraint " + cn.str()).link,
                                 km.has("body12joint1_pose") ? km["body12joint1_pose"].pose() : IdentityPose(),
                                 km.has("body22joint2_pose") ? km["body22joint2_pose"].pose() : IdentityPose(), directions(km));
    }
    for (auto& cn : (*n)["soft_constraints"].seq) {
      const Node& k = *detail::Find(soft_constraint_nodes, cn.str(), by);
      Node km = k.has("metafile_path") ? ReadYaml(meta(k)) : Node{};
      auto number = [&](const char* key, float fallback) { return km.has(key) ? float(km[key].number()) : fallback; };
      o.optimizer->AddSoftConstraint(*detail::Find(t->links, k["link1"].str(), "SoftConstraint " + cn.str()).link,
                                     *detail::Find(t->links, k["link2"].str(), "SoftConstraint " + cn.str()).link,
                                     km.has("body12joint1_pose") ? km["body12joint1_pose"].pose() : IdentityPose(),
                                     km.has("body22joint2_pose") ? km["body22joint2_pose"].pose() : IdentityPose(),
                                     directions(km), number("max_distance_rotation", 0.0f),
                                     number("max_distance_translation", 0.0f), number("standard_deviation_rotation", 0.01f),
                                     number("standard_deviation_translation", 0.001f));
    }
    t->optimizers[name] = o;
  }
  for (auto* n : detail::Entries(root, "StaticDetector", {"name", "metafile_path", "optimizer"}, path)) {
    Node m = ReadYaml(meta(*n));
    Required(m, {"link2world_pose"}, "static detector", meta(*n));
    StaticDetector d;
    d.name = (*n)["name"].str();
    d.optimizer_name = (*n)["optimizer"].str();
    const ConfiguredOptimizer& o = detail::Find(t->optimizers, d.optimizer_name, "StaticDetector " + d.name);
    d.root_link = o.root_link;
    d.root_link_name = o.root_link_name;
    d.link2world_pose = m["link2world_pose"].pose();
    if (m.has("reset_joint_poses")) d.reset_joint_poses = m["reset_joint_poses"].boolean();
    t->detectors.push_back(d);
  }
  std::map<std::string, ConfiguredRefiner> refiners;
  for (auto* n : detail::Entries(root, "Refiner", {"name", "optimizers"}, path)) {
    Node m = n->has("metafile_path") ? ReadYaml(meta(*n)) : Node{};
    ConfiguredRefiner r;
    r.name = (*n)["name"].str();
    if (m.has("n_corr_iterations")) r.n_corr_iterations = m["n_corr_iterations"].integer();
    if (m.has("n_update_iterations")) r.n_update_iterations = m["n_update_iterations"].integer();
    refiners[r.name] = r;
  }
  for (const char* k : {"ImageColorViewer", "ImageDepthViewer", "NormalColorViewer", "NormalDepthViewer"})
    for (auto& n : root[k].seq) t->ignored.push_back(n["name"].str());
  auto trackers = detail::Entries(root, "Tracker", {"name", "optimizers"}, path);
  if (trackers.empty()) throw std::runtime_error("No tracker was configured in " + path);
  if (trackers.size() > 1) throw std::runtime_error("More than one tracker was configured in " + path);
  const Node& tn = *trackers[0];
  t->name = tn["name"].str();
  Node m = tn.has("metafile_path") ? ReadYaml(meta(tn)) : Node{};
  if (m.has("n_corr_iterations")) t->n_corr_iterations = m["n_corr_iterations"].integer();
  if (m.has("n_update_iterations")) t->n_update_iterations = m["n_update_iterations"].integer();
  std::set<std::string> used;
  for (auto& o : tn["optimizers"].seq) {
    detail::Find(t->optimizers, o.str(), "Tracker " + t->name);
    used.insert(o.str());
  }
  if (used.size() != t->optimizers.size())
    throw std::runtime_error("Optimizers are configured that are not part of tracker " + t->name +
                             ": one device context runs one tracker");
  std::vector<StaticDetector> kept;
  for (auto& d : tn["detectors"].seq) {
    bool found = false;
    for (auto& have : t->detectors)
      if (have.name == d.str()) { kept.push_back(have); found = true; }
    if (!found) throw std::runtime_error("Object " + d.str() + " required by Tracker " + t->name + " was not found");
  }
  t->detectors = kept;
  for (auto& r : tn["refiners"].seq) t->refiners.push_back(detail::Find(refiners, r.str(), "Tracker " + t->name));
  t->tracker = std::make_unique<Tracker>(c, t->n_corr_iterations, t->n_update_iterations);
  return t;
}

}  // namespace config
}  // namespace m3t_hip

#endif  // M3T_HIP_CONFIG_HPP_
