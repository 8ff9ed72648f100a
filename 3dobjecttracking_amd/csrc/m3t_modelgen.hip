// m3t_modelgen.hip — sparse viewpoint model generation without OpenGL:
// RegionModel::GenerateModel / DepthModel::GenerateModel (region_model.cpp:187-258,457-795,
// depth_model.cpp:144-212,302-351, model.cpp:120-153,338-454) for one body.  The 2562 template views
// are rasterised on the device (same OpenGL rules as m3t_render.hip, 64-bit z-buffer words
// depth16 << 32 | triangle so that the flat normal of the visible triangle is known); the per-view
// sampling (cv::findContours border following, std::mt19937{7} draws, contour normals, line
// distances, depth offsets) runs on the host, a few hundred points per view.
// The algorithm is the one restated in oracle/gl_model.py (numpy checker), which reproduces the reference's
// own generated model files; tests/test_gpu_model_generation.py compares this implementation with
// those files and with gl_model.py.  Included by m3t_hip_api.hip after m3t_render.hip.
#ifndef M3T_MODELGEN_HIP_
#define M3T_MODELGEN_HIP_

#include <random>
#include <set>

struct ModelRenderDev {
  const float* vertices;
  const int* triangles;
  int n_triangles;
  int culling;
  int image_size;
  int order;                     // draw order of this body inside its renderer (GL_LESS: the body drawn first keeps a tie)
  const float* trans;            // [n_views_in_batch][16] projection * world2camera * geometry2body, column-major
  unsigned long long* z_buffer;  // [n_views_in_batch][image_size^2]: depth16 << 32 | order << 26 | triangle
};
#define M3T_MODEL_TRIANGLE_BITS 26

extern "C" {

// grid (slices, views): every workgroup rasterises a slice of the triangles of one view.  Triangles whose
// bounding box is small are finished by the thread that owns them; large ones are queued in LDS and
// rasterised by the whole workgroup.
__global__ void __launch_bounds__(M3T_BLOCK_THREADS) model_render_kernel(ModelRenderDev job) {
  constexpr int kQueue = 256;
  __shared__ int queue[kQueue];
  __shared__ int n_queued;
  const int view = blockIdx.y, S = job.image_size;
  const int tid = threadIdx.x, nt = blockDim.x;
  unsigned long long* z_buffer = job.z_buffer + (size_t)view * S * S;
  M44 trans = load44(job.trans + (size_t)view * 16);
  const float half_s = 0.5f * (float)S;
  const int per_block = (job.n_triangles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per_block, t_end = min(t_begin + per_block, job.n_triangles);

  auto setup = [&](int t, long long* ax, long long* ay, double* zs, long long& area) -> bool {
    long long sx[3], sy[3];
    float wz[3];
    bool behind = false;
    for (int k = 0; k < 3; ++k) {
      const float* p = job.vertices + (size_t)job.triangles[t * 3 + k] * 3;
      float cx = ((trans(0, 0) * p[0] + trans(0, 1) * p[1]) + trans(0, 2) * p[2]) + trans(0, 3);
      float cy = ((trans(1, 0) * p[0] + trans(1, 1) * p[1]) + trans(1, 2) * p[2]) + trans(1, 3);
      float cz = ((trans(2, 0) * p[0] + trans(2, 1) * p[1]) + trans(2, 2) * p[2]) + trans(2, 3);
      float cw = ((trans(3, 0) * p[0] + trans(3, 1) * p[1]) + trans(3, 2) * p[2]) + trans(3, 3);
      if (!(cw > 0.0f)) behind = true;
      float wx = (cx / cw + 1.0f) * half_s;
      float wy = (cy / cw + 1.0f) * half_s;
      wz[k] = (cz / cw + 1.0f) * 0.5f;
      sx[k] = (long long)floor((double)wx * 256.0 + 0.5);
      sy[k] = (long long)floor((double)wy * 256.0 + 0.5);
    }
    if (behind) return false;
    area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sy[1] - sy[0]) * (sx[2] - sx[0]);
    if (area == 0) return false;
    if (area > 0 && job.culling) return false;
    int i1 = 1, i2 = 2;
    if (area < 0) { i1 = 2; i2 = 1; area = -area; }
    ax[0] = sx[0]; ax[1] = sx[i1]; ax[2] = sx[i2];
    ay[0] = sy[0]; ay[1] = sy[i1]; ay[2] = sy[i2];
    zs[0] = (double)wz[0]; zs[1] = (double)wz[i1]; zs[2] = (double)wz[i2];
    return true;
  };
  auto bbox = [&](const long long* ax, const long long* ay, int& x0, int& x1, int& y0, int& y1) {
    const long long min_x = min(ax[0], min(ax[1], ax[2])), max_x = max(ax[0], max(ax[1], ax[2]));
    const long long min_y = min(ay[0], min(ay[1], ay[2])), max_y = max(ay[0], max(ay[1], ay[2]));
    // pixels whose centre (256 p + 128) lies inside the bounding box: nothing else can pass the edge tests
    x0 = (int)max(floor_div256(min_x - 128 + 255), 0LL);
    x1 = (int)min(floor_div256(max_x - 128), (long long)(S - 1));
    y0 = (int)max(floor_div256(min_y - 128 + 255), 0LL);
    y1 = (int)min(floor_div256(max_y - 128), (long long)(S - 1));
  };
  auto shade = [&](int t, const long long* ax, const long long* ay, const double* zs, double a2, int px, int py) {
    const long long cx = (long long)px * 256 + 128, cy = (long long)py * 256 + 128;
    long long e[3];
    bool inside = true;
    for (int k = 0; k < 3; ++k) {
      const int k1 = (k + 1) % 3;
      const long long dx = ax[k1] - ax[k], dy = ay[k1] - ay[k];
      e[k] = dx * (cy - ay[k]) - dy * (cx - ax[k]);
      const bool owns = dy < 0 || (dy == 0 && dx > 0);
      inside = inside && (e[k] > 0 || (e[k] == 0 && owns));
    }
    if (!inside) return;
    const double z = ((double)e[1] / a2) * zs[0] + ((double)e[2] / a2) * zs[1] + ((double)e[0] / a2) * zs[2];
    if (!(z >= 0.0 && z <= 1.0)) return;
    const unsigned long long d16 = (unsigned long long)floor(z * 65535.0 + 0.46);
    atomicMin(&z_buffer[(size_t)py * S + px],
              (d16 << 32) | ((unsigned long long)job.order << M3T_MODEL_TRIANGLE_BITS) | (unsigned long long)(unsigned)t);
  };

  if (tid == 0) n_queued = 0;
  __syncthreads();
  for (int base = t_begin; base < t_end; base += nt * 0 + nt) {
    const int t = base + tid;
    if (t < t_end) {
      long long ax[3], ay[3], area;
      double zs[3];
      if (setup(t, ax, ay, zs, area)) {
        int x0, x1, y0, y1;
        bbox(ax, ay, x0, x1, y0, y1);
        const long long pixels = (long long)max(x1 - x0 + 1, 0) * max(y1 - y0 + 1, 0);
        if (pixels > 1024) {
          int slot = atomicAdd(&n_queued, 1);
          if (slot < kQueue) queue[slot] = t;
          else {  // queue full: rasterise alone (rare)
            const double a2 = (double)area;
            for (int py = y0; py <= y1; ++py)
              for (int px = x0; px <= x1; ++px) shade(t, ax, ay, zs, a2, px, py);
          }
        } else {
          const double a2 = (double)area;
          for (int py = y0; py <= y1; ++py)
            for (int px = x0; px <= x1; ++px) shade(t, ax, ay, zs, a2, px, py);
        }
      }
    }
    __syncthreads();
    const int nq = min(n_queued, kQueue);
    for (int q = 0; q < nq; ++q) {  // large triangles: the whole workgroup strides over the bounding box
      const int tq = queue[q];
      long long ax[3], ay[3], area;
      double zs[3];
      setup(tq, ax, ay, zs, area);
      int x0, x1, y0, y1;
      bbox(ax, ay, x0, x1, y0, y1);
      const int w = x1 - x0 + 1;
      const long long total = (long long)w * (y1 - y0 + 1);
      const double a2 = (double)area;
      for (long long i = tid; i < total; i += nt) shade(tq, ax, ay, zs, a2, x0 + (int)(i % w), y0 + (int)(i / w));
    }
    __syncthreads();
    if (tid == 0) n_queued = 0;
    __syncthreads();
  }
}

// depth16 (65535 = nothing), the body that owns the pixel (its draw order, 255 = nothing) and, for the depth model, the
// visible triangle: 1 to 7 bytes per pixel cross PCIe instead of the 8-byte z-buffer words (any output may be null)
__global__ void __launch_bounds__(M3T_BLOCK_THREADS)
model_unpack_kernel(const unsigned long long* z_buffer, size_t n, uint16_t* depth, int* triangle, uint8_t* body) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long v = z_buffer[i];
    const bool hit = v != ~0ull;
    if (depth) depth[i] = hit ? (uint16_t)(v >> 32) : (uint16_t)65535;
    if (triangle) triangle[i] = hit ? (int)(v & ((1ull << M3T_MODEL_TRIANGLE_BITS) - 1)) : -1;
    if (body) body[i] = hit ? (uint8_t)((v >> M3T_MODEL_TRIANGLE_BITS) & 63ull) : (uint8_t)255;
  }
}

}  // extern "C"

namespace modelgen {

struct P4 {  // 4x4 pose, column-major, float arithmetic in the order the oracle uses
  float m[16];
  float operator()(int r, int c) const { return m[c * 4 + r]; }
  float& operator()(int r, int c) { return m[c * 4 + r]; }
};
inline P4 Identity() {
  P4 p;
  for (int i = 0; i < 16; ++i) p.m[i] = (i % 5 == 0) ? 1.0f : 0.0f;
  return p;
}
inline P4 MulAffine(const P4& a, const P4& b) {  // Transform * Transform
  P4 r = Identity();
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 3; ++k) r(k, c) = (a(k, 0) * b(0, c) + a(k, 1) * b(1, c)) + a(k, 2) * b(2, c);
  for (int k = 0; k < 3; ++k) r(k, 3) = ((a(k, 0) * b(0, 3) + a(k, 1) * b(1, 3)) + a(k, 2) * b(2, 3)) + a(k, 3);
  return r;
}
inline P4 MulGeneral(const P4& a, const P4& b) {
  P4 o;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) o(r, c) = ((a(r, 0) * b(0, c) + a(r, 1) * b(1, c)) + a(r, 2) * b(2, c)) + a(r, 3) * b(3, c);
  return o;
}
inline P4 InverseAffine(const P4& t) {  // Transform::inverse(Affine): cofactor inverse of the linear part
  auto cof = [&](int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return t(i1, j1) * t(i2, j2) - t(i1, j2) * t(i2, j1);
  };
  float c00 = cof(0, 0), c10 = cof(1, 0), c20 = cof(2, 0);
  float det = (c00 * t(0, 0) + c10 * t(1, 0)) + c20 * t(2, 0);
  float invdet = 1.0f / det;
  P4 r = Identity();
  for (int rr = 0; rr < 3; ++rr)
    for (int cc = 0; cc < 3; ++cc) r(rr, cc) = cof(cc, rr) * invdet;
  for (int k = 0; k < 3; ++k) r(k, 3) = -((r(k, 0) * t(0, 3) + r(k, 1) * t(1, 3)) + r(k, 2) * t(2, 3));
  return r;
}
inline void Apply(const P4& t, const float v[3], float out[3]) {
  for (int k = 0; k < 3; ++k) out[k] = t(k, 3) + ((t(k, 0) * v[0] + t(k, 1) * v[1]) + t(k, 2) * v[2]);
}

struct V3 { float x, y, z; };
struct V3Less {  // Model::CompareSmallerVector3f model.h:62-68
  bool operator()(const V3& a, const V3& b) const {
    return a.x < b.x || (a.x == b.x && a.y < b.y) || (a.x == b.x && a.y == b.y && a.z < b.z);
  }
};
inline V3 Normalized(V3 v) {  // Eigen: x^2 + (y^2 + z^2)
  float n = std::sqrt(v.x * v.x + (v.y * v.y + v.z * v.z));
  return {v.x / n, v.y / n, v.z / n};
}
inline void Subdivide(V3 v1, V3 v2, V3 v3, int n, std::set<V3, V3Less>* out) {  // model.cpp:436-454
  if (n == 0) {
    out->insert(v1); out->insert(v2); out->insert(v3);
    return;
  }
  V3 v12 = Normalized({v1.x + v2.x, v1.y + v2.y, v1.z + v2.z});
  V3 v13 = Normalized({v1.x + v3.x, v1.y + v3.y, v1.z + v3.z});
  V3 v23 = Normalized({v2.x + v3.x, v2.y + v3.y, v2.z + v3.z});
  Subdivide(v1, v12, v13, n - 1, out);
  Subdivide(v2, v12, v23, n - 1, out);
  Subdivide(v3, v13, v23, n - 1, out);
  Subdivide(v12, v13, v23, n - 1, out);
}
// Model::GenerateGeodesicPoses model.cpp:386-434
inline std::vector<P4> GeodesicPoses(int n_divides, float sphere_radius) {
  constexpr float x = 0.525731112119133606f, z = 0.850650808352039932f;
  const V3 ico[12] = {{-x, 0, z}, {x, 0, z}, {-x, 0, -z}, {x, 0, -z}, {0, z, x}, {0, z, -x},
                      {0, -z, x}, {0, -z, -x}, {z, x, 0}, {-z, x, 0}, {z, -x, 0}, {-z, -x, 0}};
  const int ids[20][3] = {{0, 4, 1}, {0, 9, 4}, {9, 5, 4}, {4, 5, 8}, {4, 8, 1}, {8, 10, 1}, {8, 3, 10},
                          {5, 3, 8}, {5, 2, 3}, {2, 7, 3}, {7, 10, 3}, {7, 6, 10}, {7, 11, 6}, {11, 0, 6},
                          {0, 1, 6}, {6, 1, 10}, {9, 0, 11}, {9, 11, 2}, {9, 2, 5}, {7, 2, 11}};
  std::set<V3, V3Less> points;
  for (auto& id : ids) Subdivide(ico[id[0]], ico[id[1]], ico[id[2]], n_divides, &points);
  std::vector<P4> poses;
  for (const V3& p : points) {
    P4 pose = Identity();
    V3 c2{-p.x, -p.y, -p.z}, c0;
    if (p.x == 0.0f && p.z == 0.0f) c0 = {1, 0, 0};
    else c0 = Normalized({c2.z, 0.0f, -c2.x});  // (0,1,0) x (-p)
    V3 c1{c2.y * c0.z - c2.z * c0.y, c2.z * c0.x - c2.x * c0.z, c2.x * c0.y - c2.y * c0.x};
    pose(0, 0) = c0.x; pose(1, 0) = c0.y; pose(2, 0) = c0.z;
    pose(0, 1) = c1.x; pose(1, 1) = c1.y; pose(2, 1) = c1.z;
    pose(0, 2) = c2.x; pose(1, 2) = c2.y; pose(2, 2) = c2.z;
    pose(0, 3) = p.x * sphere_radius; pose(1, 3) = p.y * sphere_radius; pose(2, 3) = p.z * sphere_radius;
    poses.push_back(pose);
  }
  return poses;
}

struct View {  // one rendered template view on the host
  int S = 0;
  float fu = 0, pp = 0, term_a = 0, term_b = 0;
  const uint16_t* depth = nullptr;  // [S*S], 65535 = nothing (a body never reaches the far plane z_max)
  const int* triangle = nullptr;    // [S*S] visible triangle or -1; only for the depth model
  // silhouette images (body ids) when the model has associated / occlusion bodies, else null:
  // main renderer (main body 255, fixed bodies 120) and the renderers of region_model.cpp:417-463 / depth_model.cpp:170-177
  const uint8_t* main_id = nullptr;
  const uint8_t *occlusion = nullptr, *same_region = nullptr, *foreground = nullptr, *background = nullptr;
  static uint8_t At(const uint8_t* image, int S, int x, int y) {
    return (x < 0 || y < 0 || x >= S || y >= S) ? 0 : image[size_t(y) * S + x];
  }
  // the main body shows at this pixel of the main renderer
  bool Covered(int x, int y) const { return main_id ? main_id[size_t(y) * S + x] == 255 : depth[size_t(y) * S + x] != 65535; }
  bool Foreground(int x, int y) const { return foreground ? foreground[size_t(y) * S + x] == 255 : Covered(x, y); }
  bool Background(int x, int y) const { return background ? background[size_t(y) * S + x] == 255 : Covered(x, y); }
  float DepthOf(uint16_t v) const { return term_a / (term_b - float(v)); }
  float DepthAt(int x, int y) const { return DepthOf(depth[size_t(y) * S + x]); }
  void PointVector(int x, int y, float out[3]) const {  // FullDepthRenderer::PointVector renderer.cpp:445-452
    float d = DepthAt(x, y);
    out[0] = d * (float(x) - pp) / fu;
    out[1] = d * (float(y) - pp) / fu;
    out[2] = d;
  }
};
struct Px { int x, y; };

// cv::findContours(RETR_LIST, CHAIN_APPROX_NONE): Suzuki-Abe border following as OpenCV's
// icvFindNextContour / icvFetchContour do it; last found contour first
inline std::vector<std::vector<Px>> FindContours(const View& v) {
  const int S = v.S, W = S + 2;
  std::vector<int> img(size_t(W) * W, 0);
  for (int y = 0; y < S; ++y)
    for (int x = 0; x < S; ++x) img[size_t(y + 1) * W + x + 1] = v.Covered(x, y) ? 1 : 0;
  static const int dx8[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dy8[8] = {0, -1, -1, -1, 0, 1, 1, 1};
  std::vector<std::vector<Px>> contours;
  int nbd = 1;
  for (int y = 1; y <= S; ++y)
    for (int x = 1; x <= S + 1; ++x) {
      const int p = img[size_t(y) * W + x], prev = img[size_t(y) * W + x - 1];
      if (p == prev) continue;
      bool is_hole;
      if (prev == 0 && p == 1) is_hole = false;
      else if (p == 0 && prev >= 1) is_hole = true;
      else continue;
      ++nbd;
      const int x0 = x - (is_hole ? 1 : 0), y0 = y;
      std::vector<Px> pts;
      int s_end = is_hole ? 0 : 4, s = s_end;
      do {
        s = (s - 1) & 7;
      } while (img[size_t(y0 + dy8[s]) * W + x0 + dx8[s]] == 0 && s != s_end);
      if (s == s_end && img[size_t(y0 + dy8[s]) * W + x0 + dx8[s]] == 0) {
        img[size_t(y0) * W + x0] = -nbd;
        pts.push_back({x0 - 1, y0 - 1});
      } else {
        const int x1 = x0 + dx8[s], y1 = y0 + dy8[s];
        int x3 = x0, y3 = y0;
        for (;;) {
          s_end = s;
          for (;;) {
            ++s;
            if (img[size_t(y3 + dy8[s & 7]) * W + x3 + dx8[s & 7]] != 0) break;
          }
          s &= 7;
          int& cell = img[size_t(y3) * W + x3];
          if ((unsigned)(s - 1) < (unsigned)s_end) cell = -nbd;
          else if (cell == 1) cell = nbd;
          pts.push_back({x3 - 1, y3 - 1});
          const int x4 = x3 + dx8[s], y4 = y3 + dy8[s];
          if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
          x3 = x4;
          y3 = y4;
          s = (s + 4) & 7;
        }
      }
      contours.push_back(std::move(pts));
    }
  std::reverse(contours.begin(), contours.end());
  return contours;
}

// Model::CalculateDepthOffsets model.cpp:338-384
inline void DepthOffsets(const View& v, int cx, int cy, float pixel_to_meter, float max_radius, float stride_m,
                         float* out) {
  int n_values = int(max_radius / stride_m + 1.0f);
  float stride = stride_m / pixel_to_meter;
  float max_diameter = 2.0f * n_values * stride;
  int image_stride = int(stride + 1.0f);
  int n_image_strides = int(max_diameter / image_stride + 1.0f);
  int image_diameter = n_image_strides * image_stride;
  int radius_minus = image_diameter / 2;
  int radius_plus = image_diameter - radius_minus;
  int v_min = std::max(cy - radius_minus, 0), v_max = std::min(cy + radius_plus, v.S - 1);
  int u_min = std::max(cx - radius_minus, 0), u_max = std::min(cx + radius_plus, v.S - 1);
  uint16_t min_values[M3T_N_DEPTH_OFFSETS];
  for (auto& m : min_values) m = 65535;
  min_values[0] = v.depth[size_t(cy) * v.S + cx];
  for (int y = v_min; y <= v_max; y += image_stride)
    for (int x = u_min; x <= u_max; x += image_stride) {
      float distance = std::sqrt(float((x - cx) * (x - cx) + (y - cy) * (y - cy)));
      int i = int(distance / stride);
      if (i < n_values) min_values[i] = std::min(min_values[i], v.depth[size_t(y) * v.S + x]);
    }
  float depth_center = v.DepthAt(cx, cy);
  out[0] = depth_center - v.DepthOf(min_values[0]);
  for (int i = 1; i < M3T_N_DEPTH_OFFSETS; ++i) {
    min_values[i] = std::min(min_values[i], min_values[i - 1]);
    out[i] = depth_center - v.DepthOf(min_values[i]);
  }
}

// RegionModel::GeneratePointData region_model.cpp:457-555 for a single body
inline void RegionViewData(const View& v, const P4& camera2body, float sphere_radius, int n_points, float max_radius,
                           float stride_m, float* points /*[n_points][38]*/, float* contour_length) {
  std::fill(points, points + size_t(n_points) * M3T_REGION_POINT_FLOATS, 0.0f);
  auto contours = FindContours(v);
  contours.erase(std::remove_if(contours.begin(), contours.end(), [](const std::vector<Px>& c) { return c.size() < 15; }),
                 contours.end());  // kMinContourLength
  const float pixel_to_meter = sphere_radius / v.fu;
  const float max_depth_difference = pixel_to_meter * 10.0f;  // kMaxSurfaceGradient
  auto contour_point_valid = [&](const Px& p) {  // IsContourPointValid :598-640
    const Px neighbours[4] = {{p.x, p.y + 1}, {p.x, p.y - 1}, {p.x + 1, p.y}, {p.x - 1, p.y}};
    if (v.same_region)
      for (const Px& n : neighbours)
        if (View::At(v.same_region, v.S, n.x, n.y) != 0) return false;
    if (v.occlusion && View::At(v.occlusion, v.S, p.x, p.y) != 0) return false;
    if (v.main_id) {
      float sum = 0.0f;
      int count = 0;
      for (const Px& n : neighbours)
        if (View::At(v.main_id, v.S, n.x, n.y) == 120) {  // kDifferentBodyID
          sum += v.DepthAt(n.x, n.y);
          ++count;
        }
      if (count > 0 && sum / float(count) < v.DepthAt(p.x, p.y) - max_depth_difference) return false;
    }
    return true;
  };
  std::vector<Px> valid;
  for (auto& c : contours)
    for (auto& p : c)
      if (!v.main_id || contour_point_valid(p)) valid.push_back(p);
  *contour_length = float(valid.size()) * pixel_to_meter;
  if (valid.empty()) return;
  auto closest = [&](float u, float vv, int* cu, int* cv) {  // FindClosestContourPoint :775-790
    float best = std::numeric_limits<float>::max();
    for (auto& c : contours)
      for (auto& p : c) {
        float d = hypotf(float(p.x) - u, float(p.y) - vv);
        if (d < best) { best = d; *cu = p.x; *cv = p.y; }
      }
  };
  std::mt19937 generator{7};
  int n_tries = 0;
  for (int i = 0; i < n_points;) {
    if (n_tries++ > 100) {  // kMaxPointSamplingTries
      *contour_length = 0.0f;
      return;
    }
    const Px center = valid[int(generator() % valid.size())];
    float pc[3];
    v.PointVector(center.x, center.y, pc);
    // CalculateContourSegment :649-685
    std::vector<Px> segment;
    bool found = false;
    for (auto& c : contours) {
      for (int idx = 0; idx < int(c.size()) && !found; ++idx) {
        if (c[idx].x != center.x || c[idx].y != center.y) continue;
        int start_idx = idx - 3, end_idx = idx + 3;  // kContourNormalApproxRadius
        if (start_idx < 0) {
          segment.insert(segment.end(), c.end() + start_idx, c.end());
          start_idx = 0;
        }
        if (end_idx >= int(c.size())) {
          segment.insert(segment.end(), c.begin() + start_idx, c.end());
          start_idx = 0;
          end_idx = end_idx - int(c.size());
        }
        segment.insert(segment.end(), c.begin() + start_idx, c.begin() + end_idx + 1);
        found = true;
      }
      if (found) break;
    }
    if (!found) continue;
    if (!(hypotf(float(segment.back().x - segment.front().x), float(segment.back().y - segment.front().y)) > 3.0f))
      continue;
    float nx = -float(segment.back().y - segment.front().y), ny = float(segment.back().x - segment.front().x);
    float nn = std::sqrt(nx * nx + ny * ny);
    nx /= nn;
    ny /= nn;
    float* dp = points + size_t(i) * M3T_REGION_POINT_FLOATS;
    Apply(camera2body, pc, dp);
    for (int k = 0; k < 3; ++k) dp[3 + k] = camera2body(k, 0) * nx + camera2body(k, 1) * ny;
    const float p2m = pc[2] / v.fu;
    DepthOffsets(v, center.x, center.y, p2m, max_radius, stride_m, dp + 8);
    // CalculateLineDistances :695-770
    float u_step, v_step;
    if (std::fabs(ny) < std::fabs(nx)) {
      u_step = nx > 0.0f ? 1.0f : (nx < 0.0f ? -1.0f : 0.0f);
      v_step = ny / std::fabs(nx);
    } else {
      u_step = nx / std::fabs(ny);
      v_step = ny > 0.0f ? 1.0f : (ny < 0.0f ? -1.0f : 0.0f);
    }
    float u_in = float(center.x) + 0.5f, v_in = float(center.y) + 0.5f, u_out = u_in, v_out = v_in;
    for (;;) {
      u_in -= u_step;
      v_in -= v_step;
      const int iu = int(u_in), iv = int(v_in);
      if (iu < 0 || iu >= v.S || iv < 0 || iv >= v.S || !v.Foreground(iu, iv)) {
        int eu = center.x, ev = center.y;
        closest(u_in + u_step - 0.5f, v_in + v_step - 0.5f, &eu, &ev);
        dp[6] = p2m * hypotf(float(eu - center.x), float(ev - center.y));
        break;
      }
    }
    for (;;) {
      u_out += u_step;
      v_out += v_step;
      if (int(u_out) < 0 || int(u_out) >= v.S || int(v_out) < 0 || int(v_out) >= v.S) {
        dp[7] = std::numeric_limits<float>::max();
        break;
      }
      if (v.Background(int(u_out), int(v_out))) {
        int eu = center.x, ev = center.y;
        closest(u_out - 0.5f, v_out - 0.5f, &eu, &ev);
        dp[7] = p2m * hypotf(float(eu - center.x), float(ev - center.y));
        break;
      }
    }
    ++i;
    n_tries = 0;
  }
}

// DepthModel::GeneratePointData depth_model.cpp:302-351 (normals: the flat normal of the visible triangle through
// the RGBA8 encoding of normal_renderer.cpp:24-31,257-270)
inline void DepthViewData(const View& v, const P4& camera2body, const P4& geometry2camera,
                          const std::vector<float>& vertices, const std::vector<int>& triangles,
                          float sphere_radius, int n_points, float max_radius, float stride_m,
                          float* points /*[n_points][36]*/, float* surface_area) {
  std::fill(points, points + size_t(n_points) * M3T_DEPTH_POINT_FLOATS, 0.0f);
  // the surface is sampled where the occlusion renderer still shows the main body (depth_model.cpp:170-177,311)
  auto visible = [&](int x, int y) {
    return v.occlusion ? v.occlusion[size_t(y) * v.S + x] == 255 : v.depth[size_t(y) * v.S + x] != 65535;
  };
  size_t n_pixels = 0;
  for (int y = 0; y < v.S; ++y)
    for (int x = 0; x < v.S; ++x) n_pixels += visible(x, y) ? 1 : 0;
  const float p2m = sphere_radius / v.fu;
  *surface_area = float(n_pixels) * (p2m * p2m);
  if (n_pixels == 0) return;
  std::mt19937 generator{7};
  const unsigned total = unsigned(v.S) * unsigned(v.S);
  for (int i = 0; i < n_points; ++i) {
    int x, y;
    for (;;) {
      int idx = int(generator() % total);
      x = idx / v.S;
      y = idx % v.S;
      if (visible(x, y)) break;
    }
    float pc[3];
    v.PointVector(x, y, pc);
    float* dp = points + size_t(i) * M3T_DEPTH_POINT_FLOATS;
    Apply(camera2body, pc, dp);
    // flat normal (p2 - p1) x (p0 - p1), normalised (renderer_geometry.cpp:199-200), rotated into the camera
    const int t = v.triangle[size_t(y) * v.S + x];
    const float* p0 = &vertices[size_t(triangles[t * 3 + 0]) * 3];
    const float* p1 = &vertices[size_t(triangles[t * 3 + 1]) * 3];
    const float* p2 = &vertices[size_t(triangles[t * 3 + 2]) * 3];
    float a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, b[3] = {p0[0] - p1[0], p0[1] - p1[1], p0[2] - p1[2]};
    float n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    float nn = std::sqrt(n[0] * n[0] + (n[1] * n[1] + n[2] * n[2]));
    for (float& c : n) c /= nn;
    float decoded[3];
    for (int k = 0; k < 3; ++k) {
      float nc = (geometry2camera(k, 0) * n[0] + geometry2camera(k, 1) * n[1]) + geometry2camera(k, 2) * n[2];
      double col = std::min(std::max(0.5 - 0.5 * double(nc), 0.0), 1.0);
      uint8_t byte = uint8_t(std::floor(col * 255.0 + 0.46));
      decoded[k] = 1.0f - float(byte) / 127.5f;
    }
    for (int k = 0; k < 3; ++k)
      dp[3 + k] = (camera2body(k, 0) * decoded[0] + camera2body(k, 1) * decoded[1]) + camera2body(k, 2) * decoded[2];
    DepthOffsets(v, x, y, pc[2] / v.fu, max_radius, stride_m, dp + 6);
  }
}

}  // namespace modelgen
#endif  // M3T_MODELGEN_HIP_
